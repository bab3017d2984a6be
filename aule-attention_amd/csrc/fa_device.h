// fa_device.h -- device-side helpers shared by the gfx950 attention kernels.
//
// CDNA4 facts this file encodes (see DESIGN.md "MFMA layouts"):
//   * wavefront = 64 lanes; v_mfma_f32_32x32x16_{bf16,f16}: each lane supplies 8
//     elements of A (row i = lane&31, k-slot = (lane>>5, j)) and of B
//     (col n = lane&31, same k-slot); the 16 fp32 results per lane are
//     D[row = (r&3) + 8*(r>>2) + 4*(lane>>5)][col = lane&31], r = 0..15.
//   * ds_read_b64_tr_b16: within each 16-lane group, lane c receives for j=0..3
//     the 16-bit element (c&3) of the 8 bytes addressed by lane 4*j + (c>>2).
#pragma once
#include <hip/hip_runtime.h>

namespace aule_hip {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_t;
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((ext_vector_type(8))) short s16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2_t;

constexpr float kLog2e = 1.4426950408889634f;
constexpr float kLn2 = 0.6931471805599453f;

// Row of the 32x32 MFMA result held in accumulator register r by lane-half hi.
__device__ __forceinline__ constexpr int crow(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }

struct Bf16Traits {
    using v8 = bf16x8_t;
    static constexpr int kDType = 2;
    static __device__ __forceinline__ f32x16_t mfma(v8 a, v8 b, f32x16_t c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    }
    // round-to-nearest-even pack of two fp32 into one dword (lo = a)
    static __device__ __forceinline__ unsigned pack2(float a, float b) {
        bf16x2_t t = {(__bf16)a, (__bf16)b};
        return __builtin_bit_cast(unsigned, t);
    }
    static __device__ __forceinline__ float lo(unsigned u) { return __builtin_bit_cast(float, u << 16); }
    static __device__ __forceinline__ float hi(unsigned u) { return __builtin_bit_cast(float, u & 0xffff0000u); }
};

struct F16Traits {
    using v8 = f16x8_t;
    static constexpr int kDType = 1;
    static __device__ __forceinline__ f32x16_t mfma(v8 a, v8 b, f32x16_t c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ unsigned pack2(float a, float b) {
        f16x2_t t = {(_Float16)a, (_Float16)b};
        return __builtin_bit_cast(unsigned, t);
    }
    static __device__ __forceinline__ float lo(unsigned u) {
        f16x2_t t = __builtin_bit_cast(f16x2_t, u);
        return (float)t[0];
    }
    static __device__ __forceinline__ float hi(unsigned u) {
        f16x2_t t = __builtin_bit_cast(f16x2_t, u);
        return (float)t[1];
    }
};

template <class T>
__device__ __forceinline__ typename T::v8 as_v8(u32x4_t x) {
    return __builtin_bit_cast(typename T::v8, x);
}

template <class T>
__device__ __forceinline__ typename T::v8 as_v8(s16x4_t a, s16x4_t b) {
    s16x8_t t = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    return __builtin_bit_cast(typename T::v8, t);
}

// LDS transpose read (gfx950): 4 x 16-bit per lane, see header comment.
__device__ __forceinline__ s16x4_t lds_tr16(const char* lds_ptr) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (s16x4_t __attribute__((address_space(3)))*)(lds_ptr));
}

// v_max3_f32 without the NaN-canonicalising v_max that fmaxf() adds in front of MFMA outputs
__device__ __forceinline__ float max3(float a, float b, float c) {
#if defined(__HIP_DEVICE_COMPILE__)
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
#else
    return fmaxf(a, fmaxf(b, c));
#endif
}

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
// One rotary pair, x' = (x1 c - x2 s, x1 s + x2 c), with the contraction spelled out: rope_gfx950.hip and the forward
// kernel's fused Q rotation must round identically.
__device__ __forceinline__ void rope_pair(float x1, float x2, float c, float si, float& y1, float& y2) {
    y1 = __builtin_fmaf(x1, c, -(x2 * si));
    y2 = __builtin_fmaf(x1, si, x2 * c);
}

__device__ __forceinline__ float fast_log2(float x) { return __builtin_amdgcn_logf(x); }

// value of the other 32-lane half (lane ^ 32)
__device__ __forceinline__ float xhalf(float x) { return __shfl_xor(x, 32, 64); }
// same through v_permlane32_swap (VALU, no LDS round trip): swap(a, b) exchanges lanes 32-63 of a
// with lanes 0-31 of b; with a = b = x the two results together hold both halves' values.
__device__ __forceinline__ float xhalf_fast(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    const unsigned u = __builtin_bit_cast(unsigned, x);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    // r[0] = x[lane & 31] (low half's value everywhere), r[1] = x[32 + (lane & 31)] (high half's value)
    const unsigned other = (threadIdx.x & 32) ? r[0] : r[1];
    return __builtin_bit_cast(float, other);
#else
    return x;
#endif
}

// Work decode shared by fwd and bwd kernels: blockIdx.x -> (batch, kv head,
// q head, block index inside the sequence), heaviest causal blocks first, and
// all blocks that share one (batch, kv-head) K/V pair on one XCD (block b runs
// on XCD b % 8 on MI355X; used for L2 locality only, never for correctness).
struct WorkItem {
    int b, hk, h, blk;
};

__host__ __device__ __forceinline__ WorkItem decode_work(int bid, int B, int Hq, int Hkv, int nblk, bool heavy_first) {
    const int g = Hq / Hkv;
    const int per_unit = g * nblk;
    const int units = B * Hkv;
    int unit, within;
    if ((units & 7) == 0) {
        const int xcd = bid & 7, j = bid >> 3;
        unit = xcd + 8 * (j / per_unit);
        within = j % per_unit;
    } else {
        unit = bid / per_unit;
        within = bid % per_unit;
    }
    WorkItem w;
    w.b = unit / Hkv;
    w.hk = unit % Hkv;
    w.h = w.hk * g + (within % g);
    const int i = within / g;
    w.blk = heavy_first ? (nblk - 1 - i) : i;
    return w;
}

// The same work items in RANK order: block rank 0 of every unit (and every query head of its group), then rank 1 of every unit, ...
// (descending: rank r = block nblk - 1 - r).  For causal kernels that do not pair their blocks -- the fp32 forward and backward: a
// workgroup's time grows with its block index -- decode_work's per-unit sawtooth (64, 60, .., 4 | 64, 60, .. tiles at S = 2048) leaves
// an XCD's in-order dispatch with a long tail: list scheduling of 16 units on 32 one-workgroup CUs reaches 91 % of the ideal, on 64
// slots (two workgroups per CU) 80 %; all units' heaviest blocks first is within 1 % (round 4: tools/README.md, f32 kernels).  A unit
// still belongs to one XCD (its K / V stay in that XCD's L2).
__host__ __device__ __forceinline__ WorkItem decode_work_ranked(int bid, int B, int Hq, int Hkv, int nblk, bool descending) {
    const int g = Hq / Hkv;
    const int units = B * Hkv;
    int unit, hq, rank;
    if ((units & 7) == 0) {
        const int xcd = bid & 7, j = bid >> 3;
        const int per_rank = (units >> 3) * g;   // work items of one rank on this XCD
        rank = j / per_rank;
        const int r = j % per_rank;
        unit = xcd + 8 * (r / g);
        hq = r % g;
    } else {
        const int per_rank = units * g;
        rank = bid / per_rank;
        const int r = bid % per_rank;
        unit = r / g;
        hq = r % g;
    }
    WorkItem w;
    w.b = unit / Hkv;
    w.hk = unit % Hkv;
    w.h = w.hk * g + hq;
    w.blk = descending ? (nblk - 1 - rank) : rank;
    return w;
}

}  // namespace aule_hip
