// tools/probe_tick.hip -- is s_memtime a constant-rate counter or the shader clock?  Spin for a fixed number of
// ticks under a light load (1 wave) and a heavy load (MFMA on every SIMD) and compare with the wall clock.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
__global__ void __launch_bounds__(256) spin(float* out, unsigned long long ticks, int heavy, unsigned long long* nm) {
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), n = 0;
    f32x16 acc[4] = {};
    bf16x8 a = {}, b = {};
    while (__builtin_amdgcn_s_memtime() - t0 < ticks) {
        if (heavy) {
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[j & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[j & 3], 0, 0, 0);
            n += 16;
        }
    }
    float s = 0; for (int d = 0; d < 4; ++d) for (int r = 0; r < 16; ++r) s += acc[d][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *nm = n;
}
int main() {
    float* d; unsigned long long* nm; hipMalloc(&d, 1024 * 256 * 4); hipMalloc(&nm, 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep)
    for (int heavy = 0; heavy < 2; ++heavy) {
        unsigned long long ticks = 200000000ull;
        hipEventRecord(e0);
        spin<<<heavy ? 512 : 1, 256>>>(d, ticks, heavy, nm);
        hipEventRecord(e1); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long n; hipMemcpy(&n, nm, 8, hipMemcpyDeviceToHost);
        printf("heavy=%d: %llu ticks in %.3f ms -> %.1f ticks/us; mfma per wave %llu -> %.2f ticks/mfma (1 wave/SIMD... 2 WG/CU => x2 waves)\n", heavy, ticks, ms, ticks / (ms * 1e3), n, n ? (double)ticks / n : 0.0);
    }
    return 0;
}
