// Reader for GVDB 1.x ".vbx" volumes, as laid out by VolumeGVDB::LoadVBX/SaveVBX
// (third-party/include/gvdb/gvdb_volume_gvdb.cpp:512-690,1755-1893; node header
// gvdb_node.h:42-55; brick placement gvdb_allocator.cpp:690-700).  Only what the renderer
// needs is decoded: the level-0 node pool (brick index position + atlas offset) and channel 0
// of the atlas.  The bricks stay a LIST (position + leafdim^3 values): a sparse file with far-apart
// bricks costs memory in proportion to its bricks, never to the box they span.  Upper-level pools
// and child lists are skipped by size.  Every size read from the file is checked against the
// file's length before anything is allocated.
#pragma once
#include <string>
#include <vector>

struct VbxBricks {
    int bd = 0;                 // brick edge in voxels (leafdim, 8 for the reference's <5,5,5,4,3> trees)
    int mn[3] = { 0, 0, 0 };    // minimum brick corner, index coordinates of the file
    int dims[3] = { 0, 0, 0 };  // extents of the box spanned by the bricks, in voxels (x, y, z)
    std::vector<int> pos;       // 3 per brick: corner relative to mn (multiples of bd)
    std::vector<float> data;    // bd^3 per brick, [z][y][x]
    size_t count() const { return pos.size() / 3; }
};

bool vbx_read_bricks(const char* path, VbxBricks& out, std::string& err);

// The same volume as a dense fp32 grid [z][y][x] whose origin is the minimum brick corner (host-side helpers and
// tests); refuses boxes of more than 2^31 voxels.
bool vbx_read_dense(const char* path, std::vector<float>& dense, int& nx, int& ny, int& nz, std::string& err);
