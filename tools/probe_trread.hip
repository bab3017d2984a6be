// tools/probe_trread.hip -- does a transpose-read-fed MFMA issue slower than a ds_read_b128-fed one?  (VERDICT r2, item 2:
// in the two-waves-per-SIMD forward the PV MFMAs -- A operand from two ds_read_b64_tr_b16 -- took 55-63 cycles each against
// ~40 for the QK^T MFMAs fed by one ds_read_b128.)
//
// One MFMA stream per wave, 16 MFMAs per pass over an LDS tile, the A operand of MFMA i requested AHEAD MFMAs earlier
// (pinned with sched_group_barrier, as the production kernels pin theirs), accumulators rotating over 4 tuples.  Kinds:
//   none   operands stay in registers            b128  one ds_read_b128 per MFMA (K image: rows padded by 16 bytes)
//   tr     two ds_read_b64_tr_b16 per MFMA ([kv/4][d/16][4][16] sub-tile image of the forward's V)
// Workgroups of 256 threads (one wave per SIMD) and 512 (two); prints shader cycles per MFMA of wave 0.
//   hipcc -O3 --offload-arch=gfx950 tools/probe_trread.hip -o probe_trread && ./probe_trread
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int KIND, int AHEAD, int NT>
__global__ void __launch_bounds__(NT) k(float* out, unsigned long long* cyc, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) ((unsigned*)smem)[i] = 0x3c003c00u + (i & 255);
    __syncthreads();
    f32x16 acc[4];
    for (int d = 0; d < 4; ++d)
        for (int r = 0; r < 16; ++r) acc[d][r] = 0.f;
    bf16x8 bq;
    for (int j = 0; j < 8; ++j) bq[j] = (__bf16)(1.0f + 0.001f * lane);
    asm volatile("" : "+v"(bq));
    // b128: row lane & 31 of a [64][272]-byte image, chunk by lane half; tr: the forward's va_off
    const char* kb = smem + (lane & 31) * 272 + (lane >> 5) * 16;
    const char* tb = smem + (lane >> 5) * 8 * 128 + ((lane >> 4) & 1) * 128 + (lane & 15) * 8;
    bf16x8 op[16];
    for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 8; ++j) op[i][j] = (__bf16)(0.5f + i);
    auto rd = [&](int i) __attribute__((always_inline)) {
        if constexpr (KIND == 1) {
            const u32x4 v = *reinterpret_cast<const u32x4*>(kb + (i & 7) * 32 + (i >> 3) * 32 * 272);
            op[i] = __builtin_bit_cast(bf16x8, v);
        } else if constexpr (KIND == 2) {
            const int sk = i >> 2, d = i & 3, off = ((4 * sk) * 8 + 2 * d) * 128;
            const s16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(tb + off));
            const s16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(tb + off + 2 * 8 * 128));
            const s16x8 o = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
            op[i] = __builtin_bit_cast(bf16x8, o);
        }
    };
    constexpr int NR = KIND == 2 ? 2 : 1;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if constexpr (KIND != 0) {
#pragma unroll
            for (int i = 0; i < AHEAD; ++i) rd(i);
            __builtin_amdgcn_sched_group_barrier(0x100, NR * AHEAD, 0);
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if constexpr (KIND != 0) {
                if (i + AHEAD < 16) rd(i + AHEAD);
            }
            acc[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(op[i], bq, acc[i & 3], 0, 0, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if constexpr (KIND != 0) {
                if (i + AHEAD < 16) __builtin_amdgcn_sched_group_barrier(0x100, NR, 0);
            }
        }
        if constexpr (KIND == 0) {
            asm volatile("" : "+v"(op[0]), "+v"(op[5]), "+v"(op[10]), "+v"(op[15]));
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int d = 0; d < 4; ++d)
        for (int r = 0; r < 16; ++r) s += acc[d][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int KIND, int AHEAD, int NT>
static void run(const char* name, float* out, unsigned long long* cyc) {
    const int iters = 2000;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k<KIND, AHEAD, NT>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL((k<KIND, AHEAD, NT>), dim3(256), dim3(NT), 65536, 0, out, cyc, iters);
        hipDeviceSynchronize();
    }
    unsigned long long c = 0;
    hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-5s ahead %d  %d waves/SIMD: %6.1f cycles per MFMA (wave 0: %d MFMAs)\n", name, AHEAD, NT / 256, (double)c / (16.0 * iters), 16 * iters);
}

int main() {
    float* out;
    unsigned long long* cyc;
    hipMalloc(&out, 256 * 512 * 4);
    hipMalloc(&cyc, 64);
    run<0, 1, 256>("none", out, cyc);
    run<1, 1, 256>("b128", out, cyc);
    run<1, 2, 256>("b128", out, cyc);
    run<1, 4, 256>("b128", out, cyc);
    run<2, 1, 256>("tr", out, cyc);
    run<2, 2, 256>("tr", out, cyc);
    run<2, 4, 256>("tr", out, cyc);
    run<0, 1, 512>("none", out, cyc);
    run<1, 1, 512>("b128", out, cyc);
    run<1, 2, 512>("b128", out, cyc);
    run<1, 4, 512>("b128", out, cyc);
    run<2, 1, 512>("tr", out, cyc);
    run<2, 2, 512>("tr", out, cyc);
    run<2, 4, 512>("tr", out, cyc);
    return 0;
}
