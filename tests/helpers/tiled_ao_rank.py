"""One rank of a two-process rehearsal of the exact tiled ray-cast AO (parallel_render.TiledRenderer.render_with_ao) with REAL
collectives -- gloo, both ranks on the one GPU: every rank generates and loads only its own tile, renders, all-gathers G-buffers and
hit states, casts the AO rays against its own leaves, all-reduces (MIN) the distances and finishes.  Rank 0 also renders the
unsplit volume with the same settings and compares bit for bit.  Started by tests/test_fullsize_gpu.py (env: RANK, WORLD_SIZE,
MASTER_ADDR, MASTER_PORT)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from isosurfacesuperresolution_amd import parallel_render as PR, volumes as V      # noqa: E402
from isosurfacesuperresolution_amd.inference import DirectRenderer                    # noqa: E402

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
n, W, H, S = 128, 160, 96, 6
field = V.EjectaField(n, seed=272)


def setup(r, origin):
    for cmd, val in (("cameraLookAt", V.fmt3((0, 0, 0))), ("cameraUp", V.fmt3((0, 1, 0))), ("cameraFoV", "30.000"), ("isovalue", "0.340"),
                     ("resolution", "%d,%d" % (W, H)), ("viewport", "0,0,%d,%d" % (W, H)), ("cameraOrigin", V.fmt3(origin)),
                     ("aoradius", "0.050")):
        assert r.send_command(cmd, val) == 0


def reduce_scalar(op):
    def f(values):                                                  # numpy array in, numpy array (same dtype) out
        t = torch.from_numpy(np.ascontiguousarray(values)).clone()
        dist.all_reduce(t, op=op)
        return t.numpy().astype(values.dtype)
    return f


tiles = PR.generate_tiles(field, (world, 1, 1), ranks=[rank], reduce_max=reduce_scalar(dist.ReduceOp.MAX), reduce_min=reduce_scalar(dist.ReduceOp.MIN))
r = DirectRenderer()
o0, o1 = V.quantize3(V.orbit_camera(12)), V.quantize3(V.orbit_camera(13))
setup(r, o0)
tr = PR.TiledRenderer(r, tiles[rank])
setup(r, o1)
out = tr.render(W, H, ao_samples=S)
ok = True
if rank == 0:
    full_tiles = PR.generate_tiles(field, (world, 1, 1))
    vol = PR.assemble(full_tiles, (n, n, n))
    setup(r, o0)
    r.load_dense(vol)
    setup(r, o1)
    r.send_command("aosamples", "%d" % S)
    full = torch.empty((H, W, 12), dtype=torch.float32, device="cuda")
    r.render_direct(full)
    torch.cuda.synchronize()
    hits = int(full[..., 3].sum())
    ao = float(full[..., 10][full[..., 3] == 1].mean())
    ok = torch.equal(out, full) and hits > 2000 and 0.05 < ao < 0.999
    print("rank 0: %d hit pixels, mean AO %.3f, %d of %d values differ" % (hits, ao, int((out != full).sum()), out.numel()), flush=True)
flag = torch.tensor([1.0 if ok else 0.0])
dist.all_reduce(flag, op=dist.ReduceOp.MIN)
dist.destroy_process_group()
sys.exit(0 if flag.item() == 1.0 else 1)
