"""Writer for GVDB 1.11 ``.vbx`` volumes (the only volume format ``loadGrid`` accepts).

Layout as written by ``VolumeGVDB::SaveVBX``
(``third-party/include/gvdb/gvdb_volume_gvdb.cpp:1755-1893``) and read by ``LoadVBX``
(``:512-690``): version, four transform vectors, grid table, grid header, topology (5 levels with
log2 dims 3,4,5,5,5 = the ``<5,5,5,4,3>`` configuration of ``GPURendererDirect/Vdb2Vbx.cpp:126``),
node pools (64-byte headers, ``gvdb_node.h:42-55``), child lists (one ``uint64`` per child slot:
``grp | lev<<8 | index<<16``, all-ones = empty; ``gvdb_allocator.h:73-74``), then the float atlas in
which brick k sits at ``(k % cx, (k / cx) % cy, k / (cx*cy)) * (8 + 2*apron) + apron``
(``gvdb_allocator.cpp:690-700``).  No sample files ship with the reference, so this writer exists for
round-trip tests of ``csrc/vbx_reader.cpp`` and to convert synthetic volumes.
"""
import struct

import numpy as np

LOGDIMS = (3, 4, 5, 5, 5)
UNDEF64 = 0xFFFFFFFFFFFFFFFF


def _elem(grp, lev, idx):
    return grp | (lev << 8) | (idx << 16)


def write_vbx(path, dense, apron=1, voxelsize=(1.0, 1.0, 1.0)):
    """dense: float32 array [z][y][x]; bricks whose 8^3 voxels are all zero are not stored."""
    dense = np.ascontiguousarray(dense, dtype=np.float32)
    nz, ny, nx = dense.shape
    bz, by, bx = (nz + 7) // 8, (ny + 7) // 8, (nx + 7) // 8
    padded = np.zeros((bz * 8, by * 8, bx * 8), np.float32)
    padded[:nz, :ny, :nx] = dense
    blocks = dense_blocks_nonzero(dense, bz, by, bx)
    bricks = {(x * 8, y * 8, z * 8): padded[z * 8:z * 8 + 8, y * 8:y * 8 + 8, x * 8:x * 8 + 8]
              for z in range(bz) for y in range(by) for x in range(bx) if blocks[z, y, x]}
    return write_vbx_bricks(path, bricks, apron, voxelsize)


def write_vbx_bricks(path, bricks, apron=1, voxelsize=(1.0, 1.0, 1.0)):
    """bricks: {(x, y, z) corner in voxels, multiples of 8, >= 0: float32 [8][8][8] (z, y, x)} -- a genuinely sparse
    volume (bricks may be thousands of voxels apart; nothing dense is built).  Returns the number of bricks."""
    if not bricks:
        raise ValueError("empty volume")
    keys = sorted(bricks, key=lambda p: (p[2], p[1], p[0]))                # z-major, as a scan of a dense volume
    nb = len(keys)
    bd = 8 + 2 * apron
    cx = int(np.ceil(nb ** (1.0 / 3.0)))
    cy = cx
    cz = (nb + cx * cy - 1) // (cx * cy)
    atlas = np.zeros((cz * bd, cy * bd, cx * bd), np.float32)

    def apron_block(pos):
        """the brick with its apron layers, gathered from the (up to 26) neighbouring bricks"""
        blk = np.zeros((8 * 3, 8 * 3, 8 * 3), np.float32)
        for dz in (-1, 0, 1):
            for dy in (-1, 0, 1):
                for dx in (-1, 0, 1):
                    q = bricks.get((pos[0] + 8 * dx, pos[1] + 8 * dy, pos[2] + 8 * dz))
                    if q is not None:
                        blk[8 * (dz + 1):8 * (dz + 2), 8 * (dy + 1):8 * (dy + 2), 8 * (dx + 1):8 * (dx + 2)] = q
        return blk[8 - apron:16 + apron, 8 - apron:16 + apron, 8 - apron:16 + apron]

    ranges = [8]
    for l in range(1, 5):
        ranges.append(ranges[-1] << LOGDIMS[l])
    # nodes per level: dict pos(tuple x,y,z in voxels) -> index
    level_nodes = [dict() for _ in range(5)]
    node_records = [[] for _ in range(5)]      # (pos, value, parent_pos)
    for k, pos in enumerate(keys):
        ax, ay, az = (k % cx) * bd + apron, ((k // cx) % cy) * bd + apron, (k // (cx * cy)) * bd + apron
        atlas[az - apron:az + 8 + apron, ay - apron:ay + 8 + apron, ax - apron:ax + 8 + apron] = apron_block(pos)
        level_nodes[0][pos] = k
        node_records[0].append((pos, (ax, ay, az)))
    for l in range(1, 5):
        r = ranges[l]
        for pos in level_nodes[l - 1]:
            ppos = tuple((c // r) * r for c in pos)
            if ppos not in level_nodes[l]:
                level_nodes[l][ppos] = len(node_records[l])
                node_records[l].append((ppos, (0, 0, 0)))
    assert len(node_records[4]) == 1, "volume must fit one root node"

    def node_bytes(lev, pos, value, parent, childlist):
        return struct.pack("<BBBB3i3i3fQQQ", lev, 1, 0, 0, pos[0], pos[1], pos[2], value[0], value[1], value[2],
                           0.0, 0.0, 0.0, parent, childlist, 0)

    pools0, pools1 = [], []
    for l in range(5):
        buf0 = bytearray()
        buf1 = bytearray()
        res = 1 << LOGDIMS[l]
        for idx, (pos, value) in enumerate(node_records[l]):
            if l < 4:
                r = ranges[l + 1]
                ppos = tuple((c // r) * r for c in pos)
                parent = _elem(0, l + 1, level_nodes[l + 1][ppos])
            else:
                parent = UNDEF64
            child = _elem(1, l, idx) if l > 0 else UNDEF64
            buf0 += node_bytes(l, pos, value, parent, child)
            if l > 0:
                kids = np.full(res ** 3, UNDEF64, dtype=np.uint64)
                cr = ranges[l - 1]
                for cpos, cidx in level_nodes[l - 1].items():
                    if all((cpos[a] // ranges[l]) * ranges[l] == pos[a] for a in range(3)):
                        lx, ly, lz = ((cpos[a] - pos[a]) // cr for a in range(3))
                        kids[(lz * res + ly) * res + lx] = _elem(0, l - 1, cidx)
                buf1 += kids.tobytes()
        pools0.append(bytes(buf0))
        pools1.append(bytes(buf1))

    with open(path, "wb") as f:
        f.write(struct.pack("<BB", 1, 11))
        f.write(struct.pack("<12f", 0, 0, 0, 0, 0, 0, 1, 1, 1, 0, 0, 0))      # pretrans, angs, scale, trans
        f.write(struct.pack("<i", 1))                                          # num_grids
        f.write(struct.pack("<B", 0))                                          # no bitmasks
        table = f.tell()
        f.write(struct.pack("<Q", 0))
        goff = f.tell()
        f.write(b"\0" * 256)                                                    # grid name
        f.write(struct.pack("<BBB", ord('f'), 1, 0))
        f.write(struct.pack("<3f", *voxelsize))
        f.write(struct.pack("<i3iii", nb, 8, 8, 8, apron, 1))
        f.write(struct.pack("<Q", atlas.nbytes))
        f.write(struct.pack("<BiB", 2, 0, 0))
        f.write(struct.pack("<3i", cx, cy, cz))
        f.write(struct.pack("<3i", atlas.shape[2], atlas.shape[1], atlas.shape[0]))
        f.write(struct.pack("<iQ", 5, _elem(0, 4, 0)))
        for l in range(5):
            res = 1 << LOGDIMS[l]
            f.write(struct.pack("<9i", LOGDIMS[l], res, ranges[l], ranges[l], ranges[l],
                                len(node_records[l]), 64, len(node_records[l]) if l > 0 else 0,
                                res ** 3 * 8 if l > 0 else 0))
        for l in range(5):
            f.write(pools0[l])
        for l in range(5):
            f.write(pools1[l])
        f.write(struct.pack("<ii", 0, 4))                                       # channel type float, stride 4
        f.write(atlas.tobytes())
        f.seek(table)
        f.write(struct.pack("<Q", goff))
    return nb


def dense_blocks_nonzero(dense, bz, by, bx):
    nz, ny, nx = dense.shape
    pad = np.zeros((bz * 8, by * 8, bx * 8), dtype=bool)
    pad[:nz, :ny, :nx] = dense != 0
    return pad.reshape(bz, 8, by, 8, bx, 8).any(axis=(1, 3, 5))
