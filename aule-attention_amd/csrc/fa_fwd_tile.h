// fa_fwd_tile.h -- pieces shared by the 16-bit tiled forward kernels (fa_fwd_pp_gfx950.hip: ping-pong schedule,
// one workgroup per Q-block pair) and the one-wave-per-SIMD kernels: LDS tile
// geometry, the single-issue softmax statements and the raw buffer descriptor.  Device code only.
#pragma once
#include "fa_device.h"

namespace aule_hip {
namespace {

#ifndef AULE_MPRIO
#define AULE_MPRIO 1
#endif
#ifndef AULE_VPRIO
#define AULE_VPRIO 0
#endif
constexpr int kQBlock = 256;
constexpr int kKVTile = 64;
constexpr float kRescaleThr = 8.0f;  // lazy rescale: keep the old running max while the new one is < 2^8 larger

template <int D>
struct Cfg {
    static constexpr int RB = D * 2;            // bytes per row in global memory
    static constexpr int RBP = RB + 16;         // padded LDS row of the K tile / Q slab: each row shifts by one
                                                // 16-B slot, so a ds_read_b128 lane group (16 rows, same column)
                                                // covers 16 distinct slots, and every offset is an immediate
    static constexpr int CPR = RB / 16;         // 16-byte chunks per row
    static constexpr int KTILE = kKVTile * RBP; // bytes per K tile in LDS
    static constexpr int VTILE = kKVTile * RB;  // bytes per V tile in LDS ([kv/4][d/16][4][16] sub-tiles)
    static constexpr int NCHUNK = kKVTile * CPR;
    static constexpr int CH = (NCHUNK + 511) / 512, KS = D / 16, DB = D / 32;
    static constexpr int QSLAB = 32 * RBP;      // one wave's Q rows
    static constexpr int LDS = 2 * KTILE + 2 * VTILE + 8 * QSLAB;
    static constexpr bool kFull = (NCHUNK % 512) == 0;
};

// timeline build: make the MFMA results "used" here so that the stamp that follows is taken after them
__device__ __forceinline__ void keep_live(f32x16_t& a, f32x16_t& b) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_nop 0" : "+v"(a), "+v"(b));
#endif
}

__device__ __forceinline__ void add_pinned(float& acc, float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("v_add_f32 %0, %0, %1" : "+v"(acc) : "v"(x));
#else
    acc += x;
#endif
}

__device__ __forceinline__ float exp2_pinned(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    float r;
    asm volatile("v_exp_f32 %0, %1" : "=v"(r) : "v"(x));
    return r;
#else
    return x;
#endif
}
__device__ __forceinline__ unsigned pack_bf16_pinned(float a, float b) {
#if defined(__HIP_DEVICE_COMPILE__)
    unsigned r;
    asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
#else
    return 0;
#endif
}
__device__ __forceinline__ unsigned softmax_pair_bf16(float s0, float s1, float c, float nm, float& a0, float& a1) {
    unsigned packed = 0;
#if defined(__HIP_DEVICE_COMPILE__)
    float x0, x1;
    asm volatile(
        "v_fma_f32 %1, %5, %7, %8\n\t"
        "v_fma_f32 %2, %6, %7, %8\n\t"
        "v_exp_f32 %1, %1\n\t"
        "v_exp_f32 %2, %2\n\t"
        "v_add_f32 %3, %3, %1\n\t"
        "v_add_f32 %4, %4, %2\n\t"
        "v_cvt_pk_bf16_f32 %0, %1, %2"
        : "=v"(packed), "=&v"(x0), "=&v"(x1), "+v"(a0), "+v"(a1)
        : "v"(s0), "v"(s1), "v"(c), "v"(nm));
#endif
    return packed;
}
// Eight scores (half an S tuple) per asm statement: x = S*c - m_ref, P = exp2(x), two row-sum chains, bf16 pack.
// Inputs and temporaries are separate operands (tying them made hipcc copy the MFMA result tuple register by
// register): 8 inputs + 8 temporaries + 4 packed outputs + c, nm + the two accumulators = 24 operands (limit 30).
// Every v_exp result is first read at least two instructions later (inline asm is invisible to the hazard recogniser).
#define AULE_SM_PAIR(CVT, S0, S1, X0, X1, PK)    \
    "v_fma_f32 " X0 ", " S0 ", %22, %23\n\t"     \
    "v_fma_f32 " X1 ", " S1 ", %22, %23\n\t"     \
    "v_exp_f32 " X0 ", " X0 "\n\t"               \
    "v_exp_f32 " X1 ", " X1 "\n\t"               \
    "v_add_f32 %12, %12, " X0 "\n\t"             \
    "v_add_f32 %13, %13, " X1 "\n\t"             \
    CVT " " PK ", " X0 ", " X1 "\n\t"
#define AULE_SM_OCT(CVT)                                                                                              \
    asm volatile(AULE_SM_PAIR(CVT, "%14", "%15", "%0", "%1", "%8") AULE_SM_PAIR(CVT, "%16", "%17", "%2", "%3", "%9")    \
                 AULE_SM_PAIR(CVT, "%18", "%19", "%4", "%5", "%10") AULE_SM_PAIR(CVT, "%20", "%21", "%6", "%7", "%11")  \
                 : "=&v"(x0), "=&v"(x1), "=&v"(x2), "=&v"(x3), "=&v"(x4), "=&v"(x5), "=&v"(x6), "=&v"(x7), "=&v"(k0),   \
                   "=&v"(k1), "=&v"(k2), "=&v"(k3), "+v"(a0), "+v"(a1)                                                  \
                 : "v"(s0), "v"(s1), "v"(s2), "v"(s3), "v"(s4), "v"(s5), "v"(s6), "v"(s7), "v"(c), "v"(nm))
template <class T>
__device__ __forceinline__ u32x4_t softmax_oct(float s0, float s1, float s2, float s3, float s4, float s5, float s6, float s7,
                                                    float c, float nm, float& a0, float& a1) {
    u32x4_t pk = {0u, 0u, 0u, 0u};
#if defined(__HIP_DEVICE_COMPILE__)
    float x0, x1, x2, x3, x4, x5, x6, x7;
    unsigned k0, k1, k2, k3;
    if constexpr (T::kDType == 2) AULE_SM_OCT("v_cvt_pk_bf16_f32");
    else AULE_SM_OCT("v_cvt_pk_f16_f32");   // round-to-nearest-even, like the (_Float16) casts of F16Traits::pack2
    pk = u32x4_t{k0, k1, k2, k3};
#else
    (void)s0; (void)s1; (void)s2; (void)s3; (void)s4; (void)s5; (void)s6; (void)s7; (void)c; (void)nm; (void)a0; (void)a1;
#endif
    return pk;
}
__device__ __forceinline__ float fma_pinned(float a, float b, float c) {
#if defined(__HIP_DEVICE_COMPILE__)
    float r;
    asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
#else
    return a * b + c;
#endif
}

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_srd(const void* base, unsigned bytes) {
    // raw buffer (stride 0): loads at offsets >= bytes return 0 -> ragged tiles need no clamping
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}

}  // namespace
}  // namespace aule_hip
