// tools/probe_issue.hip -- how much VALU work of wave B issues "for free" while its SIMD partner A runs an
// MFMA + LDS-read stream (the M-phase / V-phase situation of the ping-pong attention kernels).
// 512-thread workgroups, waves 0-3 = role A (MFMA stream), waves 4-7 = role B (VALU stream of one kind).
// Prints cycles per iteration (s_memtime) for A alone, B alone and both together.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;

template <int KIND>
__device__ __forceinline__ void valu_block(float (&x)[32], float c, float m) {
    // 32 independent values; one "block" = the given op on all 32 (KIND-specific count printed by host)
    if constexpr (KIND == 0) {       // 32 v_fma_f32
#pragma unroll
        for (int i = 0; i < 32; ++i) x[i] = __builtin_fmaf(x[i], c, m);
    } else if constexpr (KIND == 1) { // 16 v_pk_fma_f32
#pragma unroll
        for (int i = 0; i < 16; ++i) { f32x2 t = {x[2*i], x[2*i+1]}; f32x2 cc = {c, c}, mm = {m, m}; t = __builtin_elementwise_fma(t, cc, mm); x[2*i] = t[0]; x[2*i+1] = t[1]; }
    } else if constexpr (KIND == 2) { // 32 v_exp_f32
#pragma unroll
        for (int i = 0; i < 32; ++i) x[i] = __builtin_amdgcn_exp2f(x[i]);
    } else if constexpr (KIND == 3) { // 16 v_cvt_pk_bf16_f32 (+16 v_lshl to keep data flowing)
#pragma unroll
        for (int i = 0; i < 16; ++i) { typedef __attribute__((ext_vector_type(2))) __bf16 b2; b2 t = {(__bf16)x[2*i], (__bf16)x[2*i+1]}; unsigned u = __builtin_bit_cast(unsigned, t); x[2*i] = __builtin_bit_cast(float, u); }
    } else if constexpr (KIND == 4) { // 16 v_pk_add_f32
#pragma unroll
        for (int i = 0; i < 16; ++i) { f32x2 t = {x[2*i], x[2*i+1]}; f32x2 mm = {m, m}; t = t + mm; x[2*i] = t[0]; x[2*i+1] = t[1]; }
    } else if constexpr (KIND == 5) { // 32 v_add_f32
#pragma unroll
        for (int i = 0; i < 32; ++i) x[i] = x[i] + m;
    }
}

template <int KIND>
__global__ void __launch_bounds__(512) k(float* out, unsigned long long* cyc, int iters, int mode, int prio) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) ((unsigned*)smem)[i] = 0x3c003c00u + i;
    __syncthreads();
    const bool roleA = wave < 4;
    const bool active = mode == 2 || (mode == 0 && roleA) || (mode == 1 && !roleA);
    float s = 0;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (active) {
        if (roleA) {
            f32x16 acc[4] = {};
            u32x4 b = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
            const char* tb = smem + lane * 8;
            if (prio) __builtin_amdgcn_s_setprio(1);
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    s16x4 x0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(tb + j * 1024 + (it & 1) * 16384));
                    s16x4 x1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(tb + j * 1024 + 512 + (it & 1) * 16384));
                    s16x8 t = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
                    acc[j & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, t), __builtin_bit_cast(bf16x8, b), acc[j & 3], 0, 0, 0);
                }
            }
            for (int d = 0; d < 4; ++d) for (int r = 0; r < 16; ++r) s += acc[d][r];
        } else {
            float x[32];
#pragma unroll
            for (int i = 0; i < 32; ++i) x[i] = 0.001f * (lane + i);
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int rep = 0; rep < 4; ++rep) valu_block<KIND>(x, 0.999f, 0.0001f);
            }
#pragma unroll
            for (int i = 0; i < 32; ++i) s += x[i];
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0 && blockIdx.x == 0) cyc[wave] = t1 - t0;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}


// every wave runs 16 MFMA + 32 tr reads + (4 valu_blocks / 2) per iteration: the same total SIMD work as mode 2,
// but interleaved inside each wave by the compiler's scheduler.
template <int KIND>
__global__ void __launch_bounds__(512) kmix(float* out, unsigned long long* cyc, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) ((unsigned*)smem)[i] = 0x3c003c00u + i;
    __syncthreads();
    float s = 0;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    f32x16 acc[4] = {};
    u32x4 b = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
    const char* tb = smem + lane * 8;
    float x[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) x[i] = 0.001f * (lane + i);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            s16x4 x0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(tb + j * 1024 + (it & 1) * 16384));
            s16x4 x1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(tb + j * 1024 + 512 + (it & 1) * 16384));
            s16x8 t = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
            acc[j & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, t), __builtin_bit_cast(bf16x8, b), acc[j & 3], 0, 0, 0);
            if ((j & 7) == 7) valu_block<KIND>(x, 0.999f, 0.0001f);
        }
    }
    for (int d = 0; d < 4; ++d) for (int r = 0; r < 16; ++r) s += acc[d][r];
#pragma unroll
    for (int i = 0; i < 32; ++i) s += x[i];
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0 && blockIdx.x == 0) cyc[wave] = t1 - t0;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int KIND>
void run(const char* name, int n_instr_per_iter) {
    float* d; unsigned long long* c; hipMalloc(&d, 256 * 512 * 4); hipMalloc(&c, 64);
    const int iters = 2000;
    unsigned long long h[8]; double res[5][2];
    for (int mode = 0; mode < 5; ++mode) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); hipEventRecord(e0);
        if (mode < 3) k<KIND><<<256, 512, 65536>>>(d, c, iters, mode, 1);
        else if (mode == 3) k<KIND><<<256, 512, 65536>>>(d, c, iters, 2, 0);
        else kmix<KIND><<<256, 512, 65536>>>(d, c, iters);
        hipEventRecord(e1); hipDeviceSynchronize();
        hipMemcpy(h, c, 64, hipMemcpyDeviceToHost);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (mode == 4) printf("   [mixed: %.1f us wall, %llu ticks -> %.0f ticks/us]\n", ms * 1e3, h[0], h[0] / (ms * 1e3));
        res[mode][0] = (double)h[0] / iters; res[mode][1] = (double)h[4] / iters;
    }
    printf("%-18s A alone %5.0f | B(%3d) alone %5.0f | prio: A %5.0f B %5.0f | noprio: A %5.0f B %5.0f | mixed-in-wave (same SIMD work) %5.0f\n", name,
           res[0][0], n_instr_per_iter, res[1][1], res[2][0], res[2][1], res[3][0], res[3][1], res[4][0]);
    hipFree(d); hipFree(c);
}
int main() {
    run<0>("128 v_fma_f32", 128); run<1>("64 v_pk_fma_f32", 64); run<2>("128 v_exp_f32", 128);
    run<3>("64 v_cvt_pk_bf16", 64); run<4>("64 v_pk_add_f32", 64); run<5>("128 v_add_f32", 128);
    return 0;
}
