// See vbx_reader.h.  Plain C++ (no HIP).
#include "vbx_reader.h"

#include <climits>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <new>

namespace {

struct File {
    FILE* fp = nullptr;
    uint64_t size = 0;
    ~File() { if (fp) std::fclose(fp); }
    bool open(const char* path)
    {
        fp = std::fopen(path, "rb");
        if (!fp) return false;
        if (fseeko(fp, 0, SEEK_END) != 0) return false;
        const off_t e = ftello(fp);
        if (e < 0 || fseeko(fp, 0, SEEK_SET) != 0) return false;
        size = (uint64_t)e;
        return true;
    }
    uint64_t left() const { const off_t p = ftello(fp); return p < 0 || (uint64_t)p > size ? 0 : size - (uint64_t)p; }
    template <typename T> bool get(T& v) { return std::fread(&v, sizeof(T), 1, fp) == 1; }
    bool get(void* p, size_t n) { return n == 0 || std::fread(p, n, 1, fp) == 1; }
    bool skip(uint64_t n) { return n <= left() && fseeko(fp, (off_t)n, SEEK_CUR) == 0; }
};

#pragma pack(push, 1)
struct VbxNode {            // gvdb_node.h:42-55, 64 bytes
    uint8_t lev, flags, priority, pad;
    int32_t pos[3];
    int32_t value[3];
    float vrange[3];
    uint64_t parent, childlist, mask;
};
#pragma pack(pop)
static_assert(sizeof(VbxNode) == 64, "GVDB node header is 64 bytes");

bool read_bricks(const char* path, VbxBricks& out, std::string& err)
{
    File f;
    if (!f.open(path)) { err = "cannot open file"; return false; }
    uint8_t major = 0, minor = 0;
    if (!f.get(major) || !f.get(minor)) { err = "truncated header"; return false; }
    if ((major == 1 && minor >= 11) || major > 1) {
        float xform[12];                       // pretrans, angs, scale, trans (:560-567)
        if (!f.get(xform, sizeof(xform))) { err = "truncated transform"; return false; }
    }
    int32_t num_grids = 0;
    if (!f.get(num_grids) || num_grids < 1 || num_grids > 1024) { err = "bad grid count"; return false; }
    uint8_t read_masks = 1;
    if ((major == 1 && minor >= 1) || major > 1) {
        if (!f.get(read_masks)) { err = "truncated header"; return false; }
    }
    std::vector<uint64_t> offs(num_grids);
    if (!f.get(offs.data(), sizeof(uint64_t) * num_grids)) { err = "truncated offsets"; return false; }

    // first grid only (the reference renders channel 0 of the first grid)
    char name[256];
    uint8_t dtype = 0, components = 0, compress = 0, topotype = 0, layout = 0;
    float voxelsize[3];
    int32_t leafcnt = 0, leafdim[3], apron = 0, num_chan = 0, reuse = 0, axiscnt[3], axisres[3];
    uint64_t atlas_sz = 0;
    if (!f.get(name, 256) || !f.get(dtype) || !f.get(components) || !f.get(compress) ||
        !f.get(voxelsize, 12) || !f.get(leafcnt) || !f.get(leafdim, 12) || !f.get(apron) ||
        !f.get(num_chan) || !f.get(atlas_sz) || !f.get(topotype) || !f.get(reuse) || !f.get(layout) ||
        !f.get(axiscnt, 12) || !f.get(axisres, 12)) { err = "truncated grid header"; return false; }
    if (compress != 0) { err = "compressed VBX not supported"; return false; }
    if (leafdim[0] <= 0 || leafdim[0] != leafdim[1] || leafdim[0] != leafdim[2] || leafdim[0] > 64 || apron < 0 || apron > 8) {
        err = "bad brick dimensions"; return false;
    }
    int32_t levels = 0;
    uint64_t root = 0;
    if (!f.get(levels) || !f.get(root) || levels < 1 || levels > 10) { err = "bad level count"; return false; }
    int32_t ld[10], res[10], range[10][3], cnt0[10], width0[10], cnt1[10], width1[10];
    uint64_t pools = 0;
    for (int n = 0; n < levels; ++n) {
        if (!f.get(ld[n]) || !f.get(res[n]) || !f.get(range[n][0]) || !f.get(range[n][1]) || !f.get(range[n][2]) ||
            !f.get(cnt0[n]) || !f.get(width0[n]) || !f.get(cnt1[n]) || !f.get(width1[n])) { err = "truncated topology"; return false; }
        if (cnt0[n] < 0 || width0[n] < 0 || cnt1[n] < 0 || width1[n] < 0) { err = "bad pool sizes"; return false; }
        pools += (uint64_t)cnt0[n] * (uint64_t)width0[n] + (uint64_t)cnt1[n] * (uint64_t)width1[n];
    }
    // nothing below is allocated before the sizes the file claims have been checked against the bytes it has
    if (pools > f.left()) { err = "node pools larger than the file"; return false; }
    if (width0[0] < (int)sizeof(VbxNode)) { err = "unsupported node width"; return false; }
    // level-0 pool: the bricks
    std::vector<VbxNode> nodes((size_t)cnt0[0]);
    {
        std::vector<uint8_t> raw((size_t)cnt0[0] * (size_t)width0[0]);
        if (!f.get(raw.data(), raw.size())) { err = "truncated node pool"; return false; }
        for (int i = 0; i < cnt0[0]; ++i) std::memcpy(&nodes[i], raw.data() + (size_t)i * width0[0], sizeof(VbxNode));
    }
    for (int n = 1; n < levels; ++n) if (!f.skip((uint64_t)cnt0[n] * (uint64_t)width0[n])) { err = "truncated pools"; return false; }
    for (int n = 0; n < levels; ++n) if (!f.skip((uint64_t)cnt1[n] * (uint64_t)width1[n])) { err = "truncated child lists"; return false; }
    if (num_chan < 1) { err = "no channels"; return false; }
    int32_t chan_type = 0, chan_stride = 0;
    if (!f.get(chan_type) || !f.get(chan_stride)) { err = "truncated channel header"; return false; }
    if (chan_stride != 4) { err = "only float channels are supported"; return false; }
    if (axisres[0] <= 0 || axisres[1] <= 0 || axisres[2] <= 0) { err = "bad atlas size"; return false; }
    const uint64_t atlas_count = (uint64_t)axisres[0] * (uint64_t)axisres[1] * (uint64_t)axisres[2];
    if (atlas_count > f.left() / sizeof(float)) { err = "atlas larger than the file"; return false; }
    std::vector<float> atlas((size_t)atlas_count);
    if (!f.get(atlas.data(), atlas.size() * sizeof(float))) { err = "truncated atlas"; return false; }

    const int bd = leafdim[0];
    int64_t mn[3] = { INT64_MAX, INT64_MAX, INT64_MAX }, mx[3] = { INT64_MIN, INT64_MIN, INT64_MIN };
    size_t used = 0;
    for (const VbxNode& nd : nodes) {
        if (!nd.flags) continue;
        ++used;
        for (int k = 0; k < 3; ++k) {
            const int64_t lo = nd.pos[k], hi = (int64_t)nd.pos[k] + bd;
            if (lo < mn[k]) mn[k] = lo;
            if (hi > mx[k]) mx[k] = hi;
        }
    }
    if (!used) { err = "volume has no bricks"; return false; }
    for (int k = 0; k < 3; ++k) {
        if (mx[k] - mn[k] > 4096) { err = "volume larger than 4096^3"; return false; }
        out.mn[k] = (int)mn[k];
        out.dims[k] = (int)(mx[k] - mn[k]);
    }
    out.bd = bd;
    const size_t per = (size_t)bd * bd * bd;
    // Every node is validated BEFORE any memory proportional to the node count is taken: atlas offsets inside the atlas, positions
    // on the brick grid, and no more bricks than the atlas has room for (many nodes may name the same atlas brick: without the cap
    // a small file could ask for thousands of times its size)
    if (used > atlas_count / per) { err = "more bricks than the atlas holds"; return false; }
    for (const VbxNode& nd : nodes) {
        if (!nd.flags) continue;
        const int64_t ax = nd.value[0], ay = nd.value[1], az = nd.value[2];
        if (ax < 0 || ay < 0 || az < 0 || ax + bd > axisres[0] || ay + bd > axisres[1] || az + bd > axisres[2]) {
            err = "brick outside atlas"; return false;
        }
        for (int a = 0; a < 3; ++a)
            if (((int64_t)nd.pos[a] - mn[a]) % bd) { err = "brick off the brick grid"; return false; }
    }
    out.pos.clear(); out.data.clear();
    out.pos.reserve(used * 3);
    out.data.resize(used * per);
    size_t k = 0;
    for (const VbxNode& nd : nodes) {
        if (!nd.flags) continue;
        const int64_t ax = nd.value[0], ay = nd.value[1], az = nd.value[2];
        if (ax < 0 || ay < 0 || az < 0 || ax + bd > axisres[0] || ay + bd > axisres[1] || az + bd > axisres[2]) {
            err = "brick outside atlas"; return false;
        }
        for (int a = 0; a < 3; ++a) {
            const int64_t rel = (int64_t)nd.pos[a] - mn[a];
            if (rel % bd) { err = "brick off the brick grid"; return false; }
            out.pos.push_back((int)rel);
        }
        float* dst = out.data.data() + k * per;
        for (int z = 0; z < bd; ++z)
            for (int y = 0; y < bd; ++y)
                std::memcpy(dst + ((size_t)z * bd + y) * bd, &atlas[((size_t)(az + z) * axisres[1] + (size_t)(ay + y)) * axisres[0] + (size_t)ax],
                            sizeof(float) * bd);
        ++k;
    }
    return true;
}

}  // namespace

bool vbx_read_bricks(const char* path, VbxBricks& out, std::string& err)
{
    try {
        return read_bricks(path, out, err);
    } catch (const std::bad_alloc&) {
        err = "out of memory";
    } catch (...) {
        err = "malformed file";
    }
    return false;
}

bool vbx_read_dense(const char* path, std::vector<float>& dense, int& nx, int& ny, int& nz, std::string& err)
{
    try {
        VbxBricks b;
        if (!read_bricks(path, b, err)) return false;
        const uint64_t count = (uint64_t)b.dims[0] * (uint64_t)b.dims[1] * (uint64_t)b.dims[2];
        if (count > (1ull << 31)) { err = "box spanned by the bricks too large for a dense copy"; return false; }
        nx = b.dims[0]; ny = b.dims[1]; nz = b.dims[2];
        dense.assign((size_t)count, 0.0f);
        const int bd = b.bd;
        const size_t per = (size_t)bd * bd * bd;
        for (size_t k = 0; k < b.count(); ++k) {
            const int px = b.pos[3 * k], py = b.pos[3 * k + 1], pz = b.pos[3 * k + 2];
            const float* src = b.data.data() + k * per;
            for (int z = 0; z < bd; ++z)
                for (int y = 0; y < bd; ++y)
                    std::memcpy(&dense[((size_t)(pz + z) * ny + (size_t)(py + y)) * nx + (size_t)px], src + ((size_t)z * bd + y) * bd, sizeof(float) * bd);
        }
        return true;
    } catch (const std::bad_alloc&) {
        err = "out of memory";
    } catch (...) {
        err = "malformed file";
    }
    return false;
}
