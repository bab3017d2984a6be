// tools/probe_mfma_lds.hip -- ceiling of "one 32x32x16 bf16 MFMA per 1 KiB of LDS operand reads" on MI355X.
// Variants: 0 = MFMA only; 1 = + one ds_read_b128 (A operand) per MFMA; 2 = + two ds_read_b64_tr_b16 per MFMA;
// each for 256 or 512 threads per workgroup (1 or 2 waves per SIMD), one workgroup per CU x 4 rounds.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

template <int VAR>
__global__ void __launch_bounds__(512) k(float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) ((unsigned*)smem)[i] = 0x3c003c00u + i;
    __syncthreads();
    f32x16 acc[4] = {};
    u32x4 b = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
    u32x4 a = b;
    const char* base = smem + (lane & 31) * 272 + (lane >> 5) * 16;
    const char* tb = smem + lane * 8;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if (VAR == 1) a = *reinterpret_cast<const u32x4*>(base + j * 32 + (it & 1) * 8704);
            if (VAR == 2) {
                s16x4 x0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(tb + j * 1024 + (it & 1) * 16384));
                s16x4 x1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(tb + j * 1024 + 512 + (it & 1) * 16384));
                s16x8 t = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
                a = __builtin_bit_cast(u32x4, t);
            }
            acc[j & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc[j & 3], 0, 0, 0);
        }
    }
    float s = 0;
    for (int d = 0; d < 4; ++d) for (int r = 0; r < 16; ++r) s += acc[d][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int VAR>
void run(int threads, const char* name) {
    float* d; hipMalloc(&d, 1024 * 512 * 4);
    const int iters = 2000, grid = 1024;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<VAR><<<grid, threads, 65536>>>(d, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<VAR><<<grid, threads, 65536>>>(d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flops = (double)grid * (threads / 64) * iters * 16 * 2.0 * 32 * 32 * 16;
    printf("%-34s threads=%d: %.3f ms  %.0f TFLOP/s\n", name, threads, ms, flops / ms / 1e9);
    hipFree(d);
}
int main() {
    run<0>(256, "MFMA only"); run<0>(512, "MFMA only");
    run<1>(256, "MFMA + ds_read_b128 each"); run<1>(512, "MFMA + ds_read_b128 each");
    run<2>(256, "MFMA + 2 ds_read_b64_tr_b16 each"); run<2>(512, "MFMA + 2 ds_read_b64_tr_b16 each");
    return 0;
}
