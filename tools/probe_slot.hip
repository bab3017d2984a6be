// tools/probe_slot.hip -- marginal cost of each instruction kind in one MFMA slot, ONE wave per SIMD
// (256-thread workgroups, 512 registers): the slot of the 4-wave attention kernels is
//   v_mfma_f32_32x32x16_bf16 + NLDS x ds_read_b64_tr_b16 (operands 2 slots ahead) + softmax step
//   {v_accvgpr_read, v_fma, v_exp, v_add, 1/2 v_cvt_pk}.
// Variants drop one kind at a time; prints shader cycles per slot.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

// bit mask of what a slot contains
enum { M_MFMA = 1, M_LDS1 = 2, M_LDS2 = 4, M_READ = 8, M_FMA = 16, M_EXP = 32, M_ADD = 64, M_CVT = 128, M_B128 = 256 };

template <int MASK>
__global__ void __launch_bounds__(256) k(float* out, unsigned long long* cyc, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) ((unsigned*)smem)[i] = 0x3c003c00u + i;
    __syncthreads();
    f32x16 acc[8];
    for (int d = 0; d < 8; ++d) for (int r = 0; r < 16; ++r) acc[d][r] = 0.f;
    f32x16 sreg;
    for (int r = 0; r < 16; ++r) sreg[r] = 0.01f * (lane + r);
    asm volatile("" : "+a"(sreg));
    const char* tb = smem + lane * 8;
    const char* kb = smem + lane * 16;
    float t[4] = {0, 0, 0, 0}, x[4] = {0, 0, 0, 0}, p[4] = {1, 1, 1, 1}, l = 0.f;
    unsigned pk = 0, pksum = 0;
    const float c = 0.127f, nm = -0.5f;
    s16x8 opnd[4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 8; ++j) opnd[i][j] = (short)(0x3c00 + i);
    bf16x8 bq;
    for (int j = 0; j < 8; ++j) bq[j] = (__bf16)1.0f;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int sl = 0; sl < 32; ++sl) {
            // operand for slot sl+2
            if constexpr (MASK & (M_LDS1 | M_LDS2)) {
                s16x4 x0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(tb + sl * 1024 + (it & 1) * 32768));
                s16x4 x1 = x0;
                if constexpr (MASK & M_LDS2) x1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(tb + sl * 1024 + 512 + (it & 1) * 32768));
                s16x8 o = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
                opnd[(sl + 2) & 3] = o;
            } else if constexpr (MASK & M_B128) {
                typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
                u32x4 v = *reinterpret_cast<const u32x4*>(kb + sl * 1024 + (it & 1) * 32768);
                opnd[(sl + 2) & 3] = __builtin_bit_cast(s16x8, v);
            }
            if constexpr (MASK & M_MFMA)
                acc[sl & 7] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, opnd[sl & 3]), bq, acc[sl & 7], 0, 0, 0);
            // softmax pipeline step (4-deep, like sp_step)
            const int a = sl & 3, b = (sl + 3) & 3, cc = (sl + 2) & 3, d = (sl + 1) & 3;
            // ONE asm statement per slot (hipcc pads separate asm statements with s_nop)
            {
                const float se = sreg[sl & 15];
                constexpr int SMK = MASK & (M_READ | M_FMA | M_EXP | M_ADD);
                const bool cv = (MASK & M_CVT) && (sl & 1);
#define ASM_OUT "=&v"(t[a]), "=&v"(x[b]), "=&v"(p[cc]), "+v"(l), "=&v"(pk)
#define ASM_IN "a"(se), "v"(t[b]), "v"(x[cc]), "v"(p[d]), "v"(p[a]), "v"(c), "v"(nm)
#define I_READ "v_accvgpr_read_b32 %0, %5\n\t"
#define I_FMA "v_fma_f32 %1, %6, %10, %11\n\t"
#define I_EXP "v_exp_f32 %2, %7\n\t"
#define I_ADD "v_add_f32 %3, %3, %8\n\t"
#define I_CVT "v_cvt_pk_bf16_f32 %4, %8, %9\n\t"
                if constexpr (SMK == (M_READ | M_FMA | M_EXP | M_ADD)) {
                    if (cv) asm volatile(I_READ I_FMA I_EXP I_ADD I_CVT : ASM_OUT : ASM_IN);
                    else asm volatile(I_READ I_FMA I_EXP I_ADD : ASM_OUT : ASM_IN);
                } else if constexpr (SMK == (M_FMA | M_EXP | M_ADD)) {
                    if (cv) asm volatile(I_FMA I_EXP I_ADD I_CVT : ASM_OUT : ASM_IN);
                    else asm volatile(I_FMA I_EXP I_ADD : ASM_OUT : ASM_IN);
                } else if constexpr (SMK == (M_READ | M_EXP | M_ADD)) {
                    if (cv) asm volatile(I_READ I_EXP I_ADD I_CVT : ASM_OUT : ASM_IN);
                    else asm volatile(I_READ I_EXP I_ADD : ASM_OUT : ASM_IN);
                } else if constexpr (SMK == (M_READ | M_FMA | M_ADD)) {
                    if (cv) asm volatile(I_READ I_FMA I_ADD I_CVT : ASM_OUT : ASM_IN);
                    else asm volatile(I_READ I_FMA I_ADD : ASM_OUT : ASM_IN);
                } else if constexpr (SMK == (M_READ | M_FMA | M_EXP)) {
                    if (cv) asm volatile(I_READ I_FMA I_EXP I_CVT : ASM_OUT : ASM_IN);
                    else asm volatile(I_READ I_FMA I_EXP : ASM_OUT : ASM_IN);
                } else if constexpr (SMK == (M_READ | M_FMA)) {
                    asm volatile(I_READ I_FMA : ASM_OUT : ASM_IN);
                } else if constexpr (SMK == M_READ) {
                    asm volatile(I_READ : ASM_OUT : ASM_IN);
                } else if constexpr (SMK == M_FMA) {
                    asm volatile(I_FMA : ASM_OUT : ASM_IN);
                } else if constexpr (SMK == M_ADD) {
                    asm volatile(I_ADD : ASM_OUT : ASM_IN);
                } else if constexpr (SMK == (M_FMA | M_ADD)) {
                    asm volatile(I_FMA I_ADD : ASM_OUT : ASM_IN);
                } else if constexpr (SMK == M_EXP) {
                    asm volatile(I_EXP : ASM_OUT : ASM_IN);
                } else if constexpr (SMK == (M_EXP | M_ADD)) {
                    asm volatile(I_EXP I_ADD : ASM_OUT : ASM_IN);
                }
                if (cv) pksum ^= pk;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = l + t[0] + x[1] + p[2] + (float)pksum;
    for (int d = 0; d < 8; ++d) for (int r = 0; r < 16; ++r) s += acc[d][r];
    if (lane == 0 && blockIdx.x == 0 && threadIdx.x < 64) cyc[0] = t1 - t0;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MASK>
void run(const char* name) {
    float* d; unsigned long long* c; hipMalloc(&d, 256 * 256 * 4); hipMalloc(&c, 64);
    const int iters = 300;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k<MASK>), hipFuncAttributeMaxDynamicSharedMemorySize, 70000);
    unsigned long long h = 0;
    for (int rep = 0; rep < 2; ++rep) {
        k<MASK><<<256, 256, 70000>>>(d, c, iters);
        hipDeviceSynchronize();
        hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
    }
    printf("%-46s %6.1f cycles/slot\n", name, (double)h / (iters * 32.0));
    hipFree(d); hipFree(c);
}

// Two waves per SIMD (512-thread workgroups, VGPR-form MFMA, S in arch VGPRs): every wave runs the whole slot
// mix itself -- {MFMA, NLDS tr reads, v_fma, v_exp, v_add, 1/2 v_cvt_pk}.  Ideal = 64 cycles per slot per wave
// (the two waves of a SIMD share its matrix pipe).
template <int NLDS, bool SOFTMAX>
__global__ void __launch_bounds__(512) k2(float* out, unsigned long long* cyc, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) ((unsigned*)smem)[i] = 0x3c003c00u + i;
    __syncthreads();
    f32x16 acc[4];
    for (int d = 0; d < 4; ++d) for (int r = 0; r < 16; ++r) acc[d][r] = 0.f;
    f32x16 sreg;
    for (int r = 0; r < 16; ++r) sreg[r] = 0.01f * (lane + r);
    asm volatile("" : "+v"(sreg));
    const char* tb = smem + lane * 8;
    float x[4] = {0, 0, 0, 0}, p[4] = {1, 1, 1, 1}, l = 0.f;
    unsigned pk = 0, pksum = 0;
    const float c = 0.127f, nm = -0.5f;
    s16x8 opnd[4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 8; ++j) opnd[i][j] = (short)(0x3c00 + i);
    bf16x8 bq;
    for (int j = 0; j < 8; ++j) bq[j] = (__bf16)1.0f;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int sl = 0; sl < 32; ++sl) {
            if constexpr (NLDS > 0) {
                s16x4 x0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(tb + sl * 1024 + (it & 1) * 32768));
                s16x4 x1 = x0;
                if constexpr (NLDS > 1) x1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(tb + sl * 1024 + 512 + (it & 1) * 32768));
                s16x8 o = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
                opnd[(sl + 2) & 3] = o;
            }
            acc[sl & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, opnd[sl & 3]), bq, acc[sl & 3], 0, 0, 0);
            if constexpr (SOFTMAX) {
                const int a = sl & 3, b = (sl + 3) & 3, cc = (sl + 2) & 3;
                const float se = sreg[sl & 15];
                if (sl & 1)
                    asm volatile("v_fma_f32 %0, %4, %7, %8\n\tv_exp_f32 %1, %5\n\tv_add_f32 %2, %2, %6\n\tv_cvt_pk_bf16_f32 %3, %6, %5"
                                 : "=&v"(x[a]), "=&v"(p[b]), "+v"(l), "=&v"(pk) : "v"(se), "v"(x[b]), "v"(p[cc]), "v"(c), "v"(nm));
                else
                    asm volatile("v_fma_f32 %0, %3, %6, %7\n\tv_exp_f32 %1, %4\n\tv_add_f32 %2, %2, %5"
                                 : "=&v"(x[a]), "=&v"(p[b]), "+v"(l) : "v"(se), "v"(x[b]), "v"(p[cc]), "v"(c), "v"(nm));
                if (sl & 1) pksum ^= pk;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = l + x[1] + p[2] + (float)pksum;
    for (int d = 0; d < 4; ++d) for (int r = 0; r < 16; ++r) s += acc[d][r];
    if (lane == 0 && blockIdx.x == 0 && threadIdx.x < 64) cyc[0] = t1 - t0;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NLDS, bool SOFTMAX>
void run2(const char* name) {
    float* d; unsigned long long* c; hipMalloc(&d, 256 * 512 * 4); hipMalloc(&c, 64);
    const int iters = 300;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k2<NLDS, SOFTMAX>), hipFuncAttributeMaxDynamicSharedMemorySize, 70000);
    unsigned long long h = 0;
    for (int rep = 0; rep < 2; ++rep) {
        k2<NLDS, SOFTMAX><<<256, 512, 70000>>>(d, c, iters);
        hipDeviceSynchronize();
        hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
    }
    printf("2 waves/SIMD: %-32s %6.1f cycles/slot/wave (ideal 64)\n", name, (double)h / (iters * 32.0));
    hipFree(d); hipFree(c);
}

int main() {
    run2<0, false>("mfma only");
    run2<2, false>("mfma + 2 tr");
    run2<0, true>("mfma + fma exp add cvt");
    run2<1, true>("mfma + 1 tr + fma exp add cvt");
    run2<2, true>("mfma + 2 tr + fma exp add cvt");
    constexpr int SM = M_READ | M_FMA | M_EXP | M_ADD | M_CVT;
    run<M_READ>("read only (no mfma)");
    run<M_FMA>("fma only");
    run<M_ADD>("add only");
    run<M_FMA | M_ADD>("fma + add");
    run<M_EXP>("exp only");
    run<M_EXP | M_ADD>("exp + add");
    run<M_READ | M_FMA>("read + fma");
    run<M_READ | M_FMA | M_EXP | M_ADD>("read fma exp add");
    run<M_LDS1 | M_LDS2>("2 tr only");
    run<M_MFMA>("mfma only");
    run<M_MFMA | M_LDS1 | M_LDS2>("mfma + 2 tr");
    run<M_MFMA | M_LDS1>("mfma + 1 tr");
    run<M_MFMA | M_B128>("mfma + 1 b128");
    run<M_MFMA | M_LDS1 | M_LDS2 | SM>("mfma + 2 tr + read fma exp add cvt (full)");
    run<M_MFMA | M_LDS1 | SM>("mfma + 1 tr + full softmax (shared operand)");
    run<M_MFMA | SM>("mfma + full softmax, no LDS");
    run<M_MFMA | M_LDS1 | M_LDS2 | (SM & ~M_READ)>("full minus read");
    run<M_MFMA | M_LDS1 | M_LDS2 | (SM & ~M_FMA)>("full minus fma");
    run<M_MFMA | M_LDS1 | M_LDS2 | (SM & ~M_EXP)>("full minus exp");
    run<M_MFMA | M_LDS1 | M_LDS2 | (SM & ~M_ADD)>("full minus add");
    run<M_MFMA | M_LDS1 | M_LDS2 | (SM & ~M_CVT)>("full minus cvt");
    run<M_LDS1 | M_LDS2 | SM>("no mfma: 2 tr + full softmax");
    run<SM>("softmax only");
    run<M_MFMA | M_READ | M_FMA>("mfma + read + fma");
    run<M_MFMA | M_EXP>("mfma + exp");
    run<M_MFMA | M_EXP | M_ADD>("mfma + exp add");
    run<M_MFMA | M_EXP | M_ADD | M_FMA | M_READ>("mfma + read fma exp add");
    run<M_MFMA | M_LDS1 | M_LDS2 | M_EXP | M_ADD | M_FMA | M_CVT>("mfma + 2tr + fma exp add cvt (VGPR-form S)");
    run<M_MFMA | M_LDS1 | M_EXP | M_ADD | M_FMA | M_CVT>("mfma + 1tr + fma exp add cvt");
    run<M_MFMA | M_B128 | SM>("mfma + 1 b128 + full softmax");
    return 0;
}
