// probe_lds_dma.hip -- what does an out-of-range lane of `buffer_load_dwordx4 ... lds` leave in LDS?
// (The forward's ragged last K / V tile relies on the answer: zeros, like an out-of-range register load.)
//   hipcc --offload-arch=gfx950 -O2 tools/probe_lds_dma.hip -o /tmp/probe_lds_dma && /tmp/probe_lds_dma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const char* g, unsigned* out, int valid_bytes) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x;
    for (int i = 0; i < 4; ++i) reinterpret_cast<unsigned*>(smem)[lane * 4 + i] = 0xffffffffu;   // NaN pattern
    __syncthreads();
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(g), 0, valid_bytes, 0x00020000);
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)smem, 16, lane * 16, 0, 0, 0);
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = 0; i < 4; ++i) out[lane * 4 + i] = reinterpret_cast<unsigned*>(smem)[lane * 4 + i];
}
int main() {
    std::vector<unsigned> h(256);
    for (int i = 0; i < 256; ++i) h[i] = 0x1000u + i;
    char* g; unsigned* o;
    hipMalloc(&g, 1024); hipMalloc(&o, 1024);
    hipMemcpy(g, h.data(), 1024, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 1024, 0, g, o, 40 * 16);   // lanes 40..63 are out of range
    std::vector<unsigned> r(256);
    hipMemcpy(r.data(), o, 1024, hipMemcpyDeviceToHost);
    int in_ok = 0, oob_zero = 0, oob_kept = 0;
    for (int l = 0; l < 64; ++l)
        for (int i = 0; i < 4; ++i) {
            const unsigned v = r[l * 4 + i];
            if (l < 40) in_ok += v == 0x1000u + l * 4 + i;
            else { oob_zero += v == 0u; oob_kept += v == 0xffffffffu; }
        }
    printf("in-range dwords correct: %d / 160; out-of-range dwords: %d zero, %d left untouched, of 96\n", in_ok, oob_zero, oob_kept);
    return 0;
}
