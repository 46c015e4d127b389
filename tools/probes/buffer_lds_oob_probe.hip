// Probe: does `buffer_load_dwordx4 ... lds` (LDS-DMA through a buffer descriptor) write ZEROS for lanes whose offset is out of the
// descriptor's range?  (If so, zero padding of a staged patch needs no zero unit and no per-lane 64-bit address.)
//   hipcc -O3 --offload-arch=gfx950 tools/probes/buffer_lds_oob_probe.hip -o /tmp/probe && /tmp/probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lvoid_t;
__global__ void k(const u32x4* src, u32x4* out, int n)
{
    extern __shared__ u32x4 lds[];
    for (int i = threadIdx.x; i < 256; i += blockDim.x) { u32x4 f = {0xdeadbeefu, 0xdeadbeefu, 0xdeadbeefu, 0xdeadbeefu}; lds[i] = f; }
    __syncthreads();
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<u32x4*>(src), 0, n * 16, 0x00020000);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // lanes 0..19: in range; 20..39: offset beyond num_records; 40..63: the BAD offset 0x80000000
    unsigned voff = lane < 20 ? lane * 3 * 16 : (lane < 40 ? (unsigned)(n + lane) * 16u : 0x80000000u);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lvoid_t*)(lds + wave * 64), 16, (int)voff, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    out[threadIdx.x] = lds[threadIdx.x];
}
int main()
{
    const int n = 100;
    u32x4 *src, *out;
    hipMalloc(&src, 4096 * 16); hipMalloc(&out, 256 * 16);
    unsigned h[4096 * 4];
    for (int i = 0; i < 4096 * 4; ++i) h[i] = 0x1000000u + i;
    hipMemcpy(src, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(256), 256 * 16, 0, src, out, n);
    unsigned o[256 * 4];
    hipMemcpy(o, out, sizeof(o), hipMemcpyDeviceToHost);
    int ok_in = 0, zero_oob = 0, zero_bad = 0, other = 0;
    for (int w = 0; w < 4; ++w)
        for (int l = 0; l < 64; ++l) {
            const unsigned* v = o + (w * 64 + l) * 4;
            if (l < 20) ok_in += (v[0] == 0x1000000u + l * 3 * 4 && v[3] == 0x1000000u + l * 3 * 4 + 3);
            else if (l < 40) { zero_oob += (v[0] == 0 && v[1] == 0 && v[2] == 0 && v[3] == 0); other += (v[0] == 0xdeadbeefu); }
            else { zero_bad += (v[0] == 0 && v[1] == 0 && v[2] == 0 && v[3] == 0); other += (v[0] == 0xdeadbeefu); }
        }
    printf("in-range lanes correct %d / 80, beyond num_records zero %d / 80, BAD offset zero %d / 96, lanes left untouched %d\n", ok_in, zero_oob, zero_bad, other);
    return 0;
}
