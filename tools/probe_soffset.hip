// tools/probe_soffset.hip -- is the scalar offset of a raw buffer access part of the range check on gfx950?
// A 4096-byte buffer of ones behind a descriptor of 1024 records; every lane loads the dword at voffset = 4 lane with
// soffset = 0 / 896 / 1024 / 4096 (the last two wholly out of range IF soffset counts).  Prints the sum over the 64 lanes.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const float* src, float* out) {
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, 1024, 0x00020000);
    const int so[4] = {0, 896, 1024, 4096};
    for (int i = 0; i < 4; ++i) {
        float v;
        const int s = __builtin_amdgcn_readfirstlane(so[i]);
        asm volatile("buffer_load_dword %0, %1, %2, %3 offen\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(threadIdx.x * 4), "s"(r), "s"(s) : "memory");
        out[i * 64 + threadIdx.x] = v;
    }
}
int main() {
    float *src, *out; hipMalloc(&src, 16384); hipMalloc(&out, 1024);
    float h[4096]; for (int i = 0; i < 4096; ++i) h[i] = 1.f;
    hipMemcpy(src, h, 16384, hipMemcpyHostToDevice);
    k<<<1, 64>>>(src, out); hipDeviceSynchronize();
    float o[256]; hipMemcpy(o, out, 1024, hipMemcpyDeviceToHost);
    const int so[4] = {0, 896, 1024, 4096};
    for (int i = 0; i < 4; ++i) { float s = 0; for (int l = 0; l < 64; ++l) s += o[i * 64 + l]; printf("soffset %4d: %2.0f of 64 lanes in range (voffset 0..252; 1024 records)\n", so[i], s); }
    return 0;
}
