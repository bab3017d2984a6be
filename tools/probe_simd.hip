// tools/probe_simd.hip -- which SIMD does wave w of a 512-thread workgroup run on? (HW_REG_HW_ID)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void __launch_bounds__(512) k(unsigned* out) {
    unsigned hwid = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);  // HW_REG_HW_ID, all 32 bits
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + (threadIdx.x >> 6)] = hwid;
}
int main() {
    unsigned* d; hipMalloc(&d, 64 * 8 * 4);
    k<<<64, 512, 65536>>>(d);
    unsigned h[512]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int b = 0; b < 12; ++b) {
        printf("wg %2d simd:", b);
        for (int w = 0; w < 8; ++w) printf(" %u", (h[b * 8 + w] >> 4) & 3);
        printf("   wave_id:");
        for (int w = 0; w < 8; ++w) printf(" %u", h[b * 8 + w] & 15);
        printf("   cu: %u se: %u\n", (h[b * 8] >> 8) & 15, (h[b * 8] >> 13) & 7);
    }
    return 0;
}
