// Reader for GVDB 1.x ".vbx" volumes, as laid out by VolumeGVDB::LoadVBX/SaveVBX
// (third-party/include/gvdb/gvdb_volume_gvdb.cpp:512-690,1755-1893; node header
// gvdb_node.h:42-55; brick placement gvdb_allocator.cpp:690-700).  Only what the renderer
// needs is decoded: the level-0 node pool (brick index position + atlas offset) and channel 0
// of the atlas; the sparse bricks are scattered into a dense fp32 grid [z][y][x] whose origin is
// the minimum brick corner.  Upper-level pools and child lists are skipped by size.
#pragma once
#include <string>
#include <vector>

bool vbx_read_dense(const char* path, std::vector<float>& dense, int& nx, int& ny, int& nz, std::string& err);
