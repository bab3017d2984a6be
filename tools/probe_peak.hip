// tools/probe_peak.hip -- wall-clock MFMA peak of v_mfma_f32_32x32x16_bf16 with 1, 2 and 4 waves per SIMD
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
template <int NT>
__global__ void __launch_bounds__(NT) kpeak(float* out, int iters, unsigned long long* cyc) {
    f32x16 acc[4];
    for (int d = 0; d < 4; ++d) for (int r = 0; r < 16; ++r) acc[d][r] = 0.f;
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(0.5f + 0.01f * (threadIdx.x & 7) + j); b[j] = (__bf16)(1.0f - 0.003f * j); }
    unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 32; ++j) acc[j & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[j & 3], 0, 0, 0);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0; for (int d = 0; d < 4; ++d) for (int r = 0; r < 16; ++r) s += acc[d][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) { cyc[0] = t1 - t0; cyc[1] = r1 - r0; }
}
int main() {
    float* d; unsigned long long* c; hipMalloc(&d, 4096 * 1024 * 4); hipMalloc(&c, 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    for (int wps = 1; wps <= 4; wps *= 2) {
        const int blocks = 256;  // one workgroup per CU, 4 * wps waves each -> wps waves per SIMD, co-resident by construction
        auto launch = [&](int n) {
            if (wps == 1) kpeak<256><<<blocks, 256>>>(d, n, c);
            else if (wps == 2) kpeak<512><<<blocks, 512>>>(d, n, c);
            else kpeak<1024><<<blocks, 1024>>>(d, n, c);
        };
        launch(100); hipDeviceSynchronize();
        hipEventRecord(e0);
        launch(iters);
        hipEventRecord(e1); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long hh[2]; hipMemcpy(hh, c, 16, hipMemcpyDeviceToHost); unsigned long long h = hh[0];
        double flops = (double)blocks * 4 * wps * iters * 32 * 32768.0;
        printf("%d wave(s)/SIMD: %.3f ms  %.0f TFLOP/s  | %.1f memtime ticks per MFMA per wave; wave 0 ran %.3f ms by s_memrealtime (100 MHz) -> memtime rate %.0f MHz\n", wps, ms, flops / ms / 1e9,
               (double)h / (iters * 32.0), hh[1] / 1e5, (double)h / (hh[1] / 100.0));
    }
    return 0;
}
