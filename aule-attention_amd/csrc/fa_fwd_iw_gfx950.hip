// fa_fwd_iw_gfx950.hip -- FlashAttention-2 forward, "in-wave ping-pong" schedule (16-bit I/O).
//
// Same arithmetic boundary, LDS tile layouts and MFMA operand maps as fa_fwd_pp_gfx950.hip; what changes
// is WHERE the two halves of the per-tile work overlap.  Measurements behind the design (tools/probe_issue.hip,
// tools/timeline.py, DESIGN.md "forward schedule"):
//
//   * plain VALU work (fma/add/max) of one wave does not run under the MFMAs of ANOTHER wave of the same
//     SIMD: 128 v_fma next to a partner's MFMA+LDS stream take 3.6x longer, whatever s_setprio says.  The
//     8-wave ping-pong kernel therefore pays max(M-phase, contended V-phase) per phase plus two
//     workgroup barriers per tile (~3900 cycles per tile against a 2048-cycle matrix floor).
//   * inside ONE wave, ~5 single-issue instructions fit in the 32-cycle shadow of each
//     v_mfma_f32_32x32x16, and v_exp / v_cvt_pk cost next to nothing there.
//
// So here a workgroup is 4 waves (one per SIMD, whole 512-entry register file), each wave owns 64 query
// rows = two 32-row blocks b0, b1, and the blocks run half a tile apart INSIDE the wave:
//
//     segment X_j :  MFMA  O1 += V_{j-1}^T P1_{j-1} ; S1_j     = K_j Q1^T        VALU softmax(S0_j) -> P0_j
//     segment Y_j :  MFMA  O0 += V_j^T P0_j         ; S0_{j+1} = K_{j+1} Q0^T    VALU softmax(S1_j) -> P1_j
//     stage V_{j+1}, K_{j+2} (global -> VGPR at the top of X_j, VGPR -> LDS after Y_j); ONE barrier per tile
//
// Every MFMA slot carries its own slice of the other block's softmax (source order = issue order, pinned
// with sched_barrier), LDS operands are requested kAhead slots early.
//
// Softmax against a STALE reference (exact algebra, different rounding): P = exp2(x - m_ref) where m_ref is
// the row maximum of the FIRST tile and is only raised when a tile's partial row sum leaves the safe range
// (> 2^kPMaxLog2, checked on the sum that is computed anyway; the rare slow path recomputes the tile from
// the retained S).  This removes the per-tile row-max chain, the cross-half exchange and the O rescale
// test from the steady state.  With PRESCALE, Q is multiplied by scale*log2(e) once (rounded back to the
// 16-bit type) and -m_ref enters through the MFMA's C operand, so x - m_ref costs no VALU at all.
#include <cstdlib>
#include <type_traits>

#include "fa_device.h"
#include "fa_kernels.h"

namespace aule_hip {
namespace {

struct FwdIWParams {
    const void* q;
    const void* k;
    const void* v;
    void* o;
    float* lse;
    int B, Hq, Hkv, Sq, Sk;
    float c;    // scale * log2(e), signed
    int nqb;    // 256-row Q blocks
    int nwork;  // work items per head: ceil(nqb/2) when pairing, else nqb
    int pair;   // process Q blocks (i, nqb-1-i) in one workgroup
    unsigned long long* dbg;  // timeline build only
};

constexpr int kIWQBlock = 256;
constexpr int kIWTile = 64;
constexpr int kIWTLMax = 512;

template <int D>
struct IWCfg {
    static constexpr int RB = D * 2;
    static constexpr int RBP = RB + 16;          // padded LDS row (K tile, epilogue slab)
    static constexpr int CPR = RB / 16;
    static constexpr int KTILE = kIWTile * RBP;
    static constexpr int VTILE = kIWTile * RB;   // [kv/4][d/16][4][16] sub-tiles
    static constexpr int NCHUNK = kIWTile * CPR;
    static constexpr int CH = NCHUNK / 256;      // 16-byte chunks per thread per tile (D=32: 1, 64: 2, 128: 4)
    static constexpr int KS = D / 16, DB = D / 32;
    static constexpr int OSLAB = 64 * RBP;       // one wave's output rows (epilogue transpose)
    static constexpr int RING = 3 * KTILE + 3 * VTILE;
    static constexpr int LDS = RING > 4 * OSLAB ? RING : 4 * OSLAB;
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t iw_srd(const void* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}

template <int V> using ic = std::integral_constant<int, V>;

// The kernel may use all 512 registers, so hipcc selects the AGPR form of every MFMA (result and C operand
// in the accumulation half of the file).  VALU instructions cannot read AGPRs: S is fetched element by
// element with v_accvgpr_read (placed by hand in the MFMA slots), and O -- which only MFMAs touch, except in
// the SAFE path below -- must never be dragged into the arch VGPRs (hipcc then moves 64 registers per
// segment back and forth), so it is rescaled in place through v_accvgpr_read/write.
__device__ __forceinline__ float acc_read(const float& a) {
#if defined(__HIP_DEVICE_COMPILE__)
    float r;
    asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(r) : "a"(a));
    return r;
#else
    return a;
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
__device__ __forceinline__ void add_pinned(float& acc, float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("v_add_f32 %0, %0, %1" : "+v"(acc) : "v"(x));
#else
    acc += x;
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
// fused per-slot softmax work of the fast path (see `slice`)
__device__ __forceinline__ float slot_first(const float& s_e) {
    float r = 0.f;
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("v_accvgpr_read_b32 %0, %1\n\ts_nop 0\n\tv_exp_f32 %0, %0" : "=v"(r) : "a"(s_e));
#endif
    return r;
}
__device__ __forceinline__ float slot_odd(const float& s_e, float& acc, float p_prev) {
    float r = 0.f;
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("v_accvgpr_read_b32 %0, %2\n\tv_add_f32 %1, %1, %3\n\tv_exp_f32 %0, %0"
                 : "=&v"(r), "+v"(acc) : "a"(s_e), "v"(p_prev));
#endif
    return r;
}
__device__ __forceinline__ float slot_even(const float& s_e, float& acc, float p_prev2, float p_prev, unsigned& packed) {
    float r = 0.f;
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("v_accvgpr_read_b32 %0, %3\n\tv_add_f32 %1, %1, %5\n\tv_exp_f32 %0, %0\n\tv_cvt_pk_bf16_f32 %2, %4, %5"
                 : "=&v"(r), "+v"(acc), "=&v"(packed) : "a"(s_e), "v"(p_prev2), "v"(p_prev));
#endif
    return r;
}
__device__ __forceinline__ void scale_acc(f32x16_t& t, float alpha) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        float tmp;
        asm volatile("v_accvgpr_read_b32 %1, %0\n\tv_mul_f32 %1, %1, %2\n\ts_nop 0\n\tv_accvgpr_write_b32 %0, %1"
                     : "+a"(t[r]), "=&v"(tmp) : "v"(alpha));
    }
#else
    (void)t; (void)alpha;
#endif
}

// SAFE = false: the hot path.  P = exp2(x) against the FIXED reference 0 (x = q~.k already carries
//   scale*log2(e) through the prescaled Q): no row maximum, no subtraction, no rescale of O -- valid while
//   every partial row sum stays inside [2^-100, 2^110], which holds for |logit * log2 e| < ~100.
// SAFE = true: classic online softmax (running maximum, O rescaled when it grows), not software-pipelined
//   into the MFMA slots.  A workgroup re-runs a Q block in this mode when the fast pass left the range.
template <class T, int D, bool CAUSAL, bool TL = false>
__global__ void __launch_bounds__(256) fa_fwd_iw_kernel(const FwdIWParams p) {
    using C = IWCfg<D>;
    using v8 = typename T::v8;
    static_assert(std::is_same<T, Bf16Traits>::value, "the fast path relies on bf16's fp32 exponent range for P");
    constexpr int RB = C::RB, RBP = C::RBP, CPR = C::CPR, KTILE = C::KTILE, VTILE = C::VTILE;
    constexpr int CH = C::CH, KS = C::KS, DB = C::DB;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const Ks = smem;
    char* const Vs = smem + 3 * KTILE;
    int* const flag = reinterpret_cast<int*>(smem + C::LDS);  // one word behind the ring / slabs

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    int tl_n = 0;
    auto stamp = [&]() __attribute__((always_inline)) {
        if constexpr (TL) {
            if (blockIdx.x == 0 && tl_n < kIWTLMax) {
                const unsigned long long t = __builtin_amdgcn_s_memtime();
                if (lane == 0) p.dbg[wave * kIWTLMax + tl_n] = t;
                ++tl_n;
            }
        }
    };

    const WorkItem w = decode_work(blockIdx.x, p.B, p.Hq, p.Hkv, p.nwork, false);
    const int Sq = p.Sq, Sk = p.Sk;
    const float c = p.c;

    const size_t kvhead = (size_t)(w.b * p.Hkv + w.hk) * Sk * RB;
    const __amdgpu_buffer_rsrc_t krs = iw_srd(reinterpret_cast<const char*>(p.k) + kvhead, (unsigned)Sk * RB);
    const __amdgpu_buffer_rsrc_t vrs = iw_srd(reinterpret_cast<const char*>(p.v) + kvhead, (unsigned)Sk * RB);

    // ---- staging maps (256 threads, CH chunks of 16 B per thread and tile; tile start in the SGPR offset).
    //      K: row-major rows padded to RBP.  V: 8 consecutive lanes fetch one [4 kv][16 d] sub-tile, so the
    //      LDS image is filled linearly by thread id.  Tiles past the end of K/V read as zeros (buffer
    //      bounds check), so staging is unconditional.
    int k_g[CH], k_lds[CH], v_g[CH];
#pragma unroll
    for (int i = 0; i < CH; ++i) {
        const int cidx = tid + 256 * i;
        const int row = cidx / CPR, cc = cidx % CPR;
        k_g[i] = row * RB + cc * 16;
        k_lds[i] = row * RBP + cc * 16;
        const int bidx = (tid >> 3) + 32 * i;  // sub-tile index = kv4 * (D/16) + d16
        v_g[i] = ((bidx / (D / 16)) * 4 + ((tid >> 1) & 3)) * RB + ((bidx % (D / 16)) * 2 + (tid & 1)) * 16;
    }
    const int ka_base = l31 * RBP + hi * 16;
    const int va_off = hi * (D / 16) * 128 + ((lane >> 4) & 1) * 128 + (lane & 15) * 8;

    u32x4_t kst[CH], vst[CH];
    auto issue_k = [&](int kv0) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < CH; ++i) kst[i] = __builtin_amdgcn_raw_buffer_load_b128(krs, k_g[i], kv0 * RB, 0);
    };
    auto issue_v = [&](int kv0) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < CH; ++i) vst[i] = __builtin_amdgcn_raw_buffer_load_b128(vrs, v_g[i], kv0 * RB, 0);
    };
    auto write_k = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < CH; ++i) *reinterpret_cast<u32x4_t*>(Ks + buf * KTILE + k_lds[i]) = kst[i];
    };
    auto write_v = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < CH; ++i) *reinterpret_cast<u32x4_t*>(Vs + buf * VTILE + tid * 16 + i * 4096) = vst[i];
    };

    if (tid == 0) *flag = 0;

    const int nparts = (p.pair && (p.nqb - 1 - w.blk) != w.blk) ? 2 : 1;
    for (int part = 0; part < nparts; ++part) {
        const int qb = p.pair ? (part == 0 ? p.nqb - 1 - w.blk : w.blk) : w.blk;
        const int q0w = qb * kIWQBlock + wave * 64;

        const int kv_hi = CAUSAL ? min(Sk, qb * kIWQBlock + kIWQBlock) : Sk;
        const int nt = (kv_hi + kIWTile - 1) / kIWTile;          // tiles staged by the workgroup (>= 1)
        const int wave_kv_hi = CAUSAL ? min(Sk, q0w + 64) : Sk;  // keys visible to this wave
        const int na = (wave_kv_hi + kIWTile - 1) / kIWTile;     // tiles this wave computes (a prefix, >= 1)

        // Q fragments (B operand of S^T = K.Q^T), prescaled by scale*log2(e) and rounded back to 16 bits
        v8 qf[2][KS];
        {
            const size_t qhead = (size_t)(w.b * p.Hq + w.h) * Sq * RB;
            const __amdgpu_buffer_rsrc_t qrs = iw_srd(reinterpret_cast<const char*>(p.q) + qhead, (unsigned)Sq * RB);
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    u32x4_t x = __builtin_amdgcn_raw_buffer_load_b128(qrs, (q0w + 32 * b + l31) * RB + (2 * ks + hi) * 16, 0, 0);
#pragma unroll
                    for (int i = 0; i < 4; ++i) x[i] = T::pack2(T::lo(x[i]) * c, T::hi(x[i]) * c);
                    qf[b][ks] = as_v8<T>(x);
                }
        }

        f32x16_t o[2][DB];
        float m[2], l[2];
        f32x16_t s[2][2];
        v8 pb[2][2][2];

        auto run_part = [&](auto safe_tag) __attribute__((always_inline)) {
            constexpr bool SAFE = decltype(safe_tag)::value != 0;
#pragma unroll
            for (int b = 0; b < 2; ++b) {
#pragma unroll
                for (int d = 0; d < DB; ++d)
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[b][d][r] = 0.f;
                m[b] = SAFE ? -INFINITY : 0.f;
                l[b] = 0.f;
            }
            // ---- prologue: K_0, V_0, K_1 -> LDS
            issue_k(0);
            issue_v(0);
            write_k(0);
            write_v(0);
            issue_k(kIWTile);
            write_k(1);
            __syncthreads();

            // One segment: the MFMA stream of block BM (PV of the tile in V buffer vbuf, then QK^T of the tile
            // in K buffer kbuf) with the softmax of block 1-BM (tile starting at key kv0) sliced into its slots.
            auto seg = [&](auto bm_tag, auto pv_tag, auto qk_tag, auto sm_tag, int vbuf, int kbuf, int kv0) __attribute__((always_inline)) {
                constexpr int BM = decltype(bm_tag)::value, BS = 1 - BM;
                constexpr bool HAS_PV = decltype(pv_tag)::value != 0, HAS_QK = decltype(qk_tag)::value != 0;
                constexpr int SM = decltype(sm_tag)::value;  // 0: no softmax, 1: plain, 2: masked (causal diagonal / ragged Sk)
                constexpr int NPV = HAS_PV ? 4 * DB : 0, NQK = HAS_QK ? 2 * KS : 0, NS = NPV + NQK;
                constexpr int kAhead = 3;
                static_assert(NS > 0, "empty segment");
                const char* vb = Vs + vbuf * VTILE + va_off;
                const char* kb = Ks + kbuf * KTILE + ka_base;
                const int qrow_s = q0w + 32 * BS + l31;  // query row of the softmaxed block

                s16x4_t a0[NS], a1[NS];
                u32x4_t kf[NS];
                auto rd = [&](int sl) __attribute__((always_inline)) {
                    if (sl < NPV) {
                        const int sk = sl / DB, d = sl % DB;
                        const int off = ((4 * sk) * (D / 16) + 2 * d) * 128;
                        a0[sl] = lds_tr16(vb + off);
                        a1[sl] = lds_tr16(vb + off + 2 * (D / 16) * 128);
                    } else {
                        const int qi = sl - NPV, ks = qi >> 1, h = qi & 1;
                        kf[sl] = *reinterpret_cast<const u32x4_t*>(kb + ks * 32 + h * 32 * RBP);
                    }
                };
                f32x16_t z;
#pragma unroll
                for (int r = 0; r < 16; ++r) z[r] = 0.f;
                auto mf = [&](int sl) __attribute__((always_inline)) {
                    if (sl < NPV) {
                        const int sk = sl / DB, d = sl % DB;
                        o[BM][d] = T::mfma(as_v8<T>(a0[sl], a1[sl]), pb[BM][sk >> 1][sk & 1], o[BM][d]);
                    } else {
                        const int qi = sl - NPV, ks = qi >> 1, h = qi & 1;
                        s[BM][h] = T::mfma(as_v8<T>(kf[sl]), qf[BM][ks], ks == 0 ? z : s[BM][h]);
                    }
                };
                // x of element e = 16 h + r of block BS's two S tuples (log2 units), masked where needed
                auto xval = [&](int e) __attribute__((always_inline)) -> float {
                    const int h = e >> 4, r = e & 15;
                    float x = acc_read(s[BS][h][r]);
                    if constexpr (SM == 2) {
                        const int kv = kv0 + 32 * h + crow(r, hi);
                        const bool vis = (kv < Sk) && (!CAUSAL || kv <= qrow_s);
                        x = vis ? x : -INFINITY;
                    }
                    return x;
                };
                // Fast-path slot work, written as asm volatile so that it STAYS in its slot (as plain IR hipcc moves
                // every exp/add/cvt behind the last MFMA of the segment): exp of element e, row-sum add of element
                // e-1, pack of (e-2, e-1).  A transcendental's result needs one independent instruction before
                // its first use, hence the one-slot lag; inline asm is invisible to the hazard recogniser.
                float acc = 0.f;
                float pe[33];
                u32x4_t pu[2][2];
                auto slice = [&](int e) __attribute__((always_inline)) {
                    if constexpr (SM == 1) {
                        // unmasked tile: ONE asm statement per element -- S_e -> VGPR, row-sum add of e-1 (also
                        // the independent instruction between the read and the exp), exp_e, pack of (e-2, e-1)
                        if (e == 0) {
                            { const float s0 = s[BS][0][0]; pe[0] = slot_first(s0); }
                        } else if (e < 32) {
                            const int h = e >> 4, r = e & 15;
                            if (e & 1) {
                                { const float se = s[BS][h][r]; pe[e] = slot_odd(se, acc, pe[e - 1]); }
                            } else {
                                const int h2 = (e - 1) >> 4, i = ((e - 1) & 15) >> 1;
                                unsigned pk;
                                const float se = s[BS][h][r];
                                pe[e] = slot_even(se, acc, pe[e - 2], pe[e - 1], pk);
                                pu[h2][i >> 2][i & 3] = pk;
                            }
                        } else {
                            add_pinned(acc, pe[31]);
                            pu[1][1][3] = pack_bf16_pinned(pe[30], pe[31]);
                        }
                        return;
                    }
                    if (e < 32) pe[e] = exp2_pinned(xval(e));
                    if (e >= 1) {
                        add_pinned(acc, pe[e - 1]);
                        if ((e - 1) & 1) {
                            const int h = (e - 1) >> 4, i = ((e - 1) & 15) >> 1;
                            pu[h][i >> 2][i & 3] = pack_bf16_pinned(pe[e - 2], pe[e - 1]);
                        }
                    }
                };
                // S of this block was written by the last MFMAs of the previous segment: leave kLead slots
                // (>= 2 MFMA issues) before the first v_accvgpr_read of it
                constexpr int kLead = NS >= 8 ? 2 : 0;
                auto first_elem = [&](int sl) __attribute__((always_inline)) { return sl <= kLead ? 0 : (32 * (sl - kLead)) / (NS - kLead); };

#pragma unroll
                for (int sl = 0; sl < kAhead && sl < NS; ++sl) rd(sl);
#pragma unroll
                for (int sl = 0; sl < NS; ++sl) {
                    if (sl + kAhead < NS) rd(sl + kAhead);
                    mf(sl);
                    if constexpr (SM != 0 && !SAFE) {
#pragma unroll
                        for (int e = first_elem(sl); e < first_elem(sl + 1); ++e) slice(e);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                if constexpr (SM != 0 && !SAFE) {
                    slice(32);
                    l[BS] += acc;
                    asm volatile("" : "+v"(pu[0][0]), "+v"(pu[0][1]), "+v"(pu[1][0]), "+v"(pu[1][1]), "+v"(l[BS]));
#pragma unroll
                    for (int h = 0; h < 2; ++h)
#pragma unroll
                        for (int kk = 0; kk < 2; ++kk) pb[BS][h][kk] = as_v8<T>(pu[h][kk]);
                }
                if constexpr (SM != 0 && SAFE) {
                    float x[32];
                    float mx = -INFINITY;
#pragma unroll
                    for (int e = 0; e < 32; ++e) {
                        x[e] = xval(e);
                        mx = fmaxf(mx, x[e]);
                    }
                    mx = fmaxf(mx, xhalf(mx));
                    const float m_new = fmaxf(m[BS], mx);
                    const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
                    const float alpha = fast_exp2(m[BS] - m_use);  // exp2(-inf) = 0 on the first tile (O = l = 0)
                    m[BS] = m_new;
                    l[BS] *= alpha;
                    if (__builtin_amdgcn_ballot_w64(alpha != 1.0f) != 0) {
#pragma unroll
                        for (int d = 0; d < DB; ++d) scale_acc(o[BS][d], alpha);
                    }
                    float a2 = 0.f;
#pragma unroll
                    for (int e = 0; e < 32; e += 2) {
                        const float p0 = fast_exp2(x[e] - m_use), p1 = fast_exp2(x[e + 1] - m_use);
                        a2 += p0 + p1;
                        const int h = e >> 4, i = (e & 15) >> 1;
                        pu[h][i >> 2][i & 3] = T::pack2(p0, p1);
                    }
                    l[BS] += a2;
#pragma unroll
                    for (int h = 0; h < 2; ++h)
#pragma unroll
                        for (int kk = 0; kk < 2; ++kk) pb[BS][h][kk] = as_v8<T>(pu[h][kk]);
                }
            };

            // pre-phase: S0_0 = K_0 Q0^T
            seg(ic<0>{}, ic<0>{}, ic<1>{}, ic<0>{}, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);

            // rolling ring indices: V_{j-1}, V_j, V_{j+1} and K_j, K_{j+1}, K_{j+2}
            int vp = 2, vc = 0, vn = 1, k0 = 0, k1 = 1, k2 = 2;
            // One tile step.  KIND (compile time): 0 first tile (of several), 1 steady state, 2 last active
            // tile, 3 the only tile, 4 post (PV of block 1's last tile), 5 idle (staging and barrier only).
            auto tile_step = [&](int j, auto kind_tag) __attribute__((always_inline)) {
                constexpr int KIND = decltype(kind_tag)::value;
                stamp();
                issue_v((j + 1) * kIWTile);
                issue_k((j + 2) * kIWTile);
                const int kv0 = j * kIWTile;
                if constexpr (KIND == 0) {
                    seg(ic<1>{}, ic<0>{}, ic<1>{}, ic<1>{}, vp, k0, kv0);
                    stamp();
                    seg(ic<0>{}, ic<1>{}, ic<1>{}, ic<1>{}, vc, k1, kv0);
                } else if constexpr (KIND == 1) {
                    seg(ic<1>{}, ic<1>{}, ic<1>{}, ic<1>{}, vp, k0, kv0);
                    stamp();
                    seg(ic<0>{}, ic<1>{}, ic<1>{}, ic<1>{}, vc, k1, kv0);
                } else if constexpr (KIND == 2) {
                    seg(ic<1>{}, ic<1>{}, ic<1>{}, ic<2>{}, vp, k0, kv0);
                    stamp();
                    seg(ic<0>{}, ic<1>{}, ic<0>{}, ic<2>{}, vc, k1, kv0);
                } else if constexpr (KIND == 3) {
                    seg(ic<1>{}, ic<0>{}, ic<1>{}, ic<2>{}, vp, k0, kv0);
                    stamp();
                    seg(ic<0>{}, ic<1>{}, ic<0>{}, ic<2>{}, vc, k1, kv0);
                } else if constexpr (KIND == 4) {
                    seg(ic<1>{}, ic<1>{}, ic<0>{}, ic<0>{}, vp, k0, kv0);
                    stamp();
                } else {
                    stamp();
                }
                stamp();
                __builtin_amdgcn_sched_barrier(0);
                write_v(vn);
                write_k(k2);
                {
                    int t = vp; vp = vc; vc = vn; vn = t;
                    t = k0; k0 = k1; k1 = k2; k2 = t;
                }
                __syncthreads();
                __builtin_amdgcn_sched_barrier(0);
                stamp();
            };
            int j = 0;
            if (na == 1) {
                tile_step(0, ic<3>{});
                j = 1;
            } else {
                tile_step(0, ic<0>{});
                for (j = 1; j + 1 < na; ++j) tile_step(j, ic<1>{});
                tile_step(j, ic<2>{});
                ++j;
            }
            if (j < nt) {
                tile_step(j, ic<4>{});
                for (++j; j < nt; ++j) tile_step(j, ic<5>{});
            } else {
                seg(ic<1>{}, ic<1>{}, ic<0>{}, ic<0>{}, vp, k0, 0);
            }
            __syncthreads();  // every wave is done with the ring
        };

        run_part(ic<0>{});
        // range check of the fast pass (NaN fails it too); one verdict per workgroup
        {
            bool ok = true;
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const float lt = l[b] + xhalf(l[b]);
                ok = ok && (lt > 0x1p-100f) && (lt < 0x1p110f);
            }
            if (__builtin_amdgcn_ballot_w64(!ok) != 0 && lane == 0) *flag = 1;
            __syncthreads();
            const int redo = *flag;
            __syncthreads();
            if (redo) {
                if (tid == 0) *flag = 0;
                run_part(ic<1>{});
            }
        }

        // ---- epilogue: O = O^T / l through a per-wave LDS slab (whole 16-byte row chunks to global);
        //      LSE = (m + log2 l) * ln2
        char* const Os = smem + wave * C::OSLAB;
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const float lt = l[b] + xhalf(l[b]);
            const float inv = 1.0f / lt;
#pragma unroll
            for (int d = 0; d < DB; ++d)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    u32x2_t u;
                    u[0] = T::pack2(o[b][d][4 * g4 + 0] * inv, o[b][d][4 * g4 + 1] * inv);
                    u[1] = T::pack2(o[b][d][4 * g4 + 2] * inv, o[b][d][4 * g4 + 3] * inv);
                    *reinterpret_cast<u32x2_t*>(Os + (32 * b + l31) * RBP + (32 * d + 8 * g4 + 4 * hi) * 2) = u;
                }
            const int qrow = q0w + 32 * b + l31;
            if (qrow < Sq && p.lse != nullptr && hi == 0)
                p.lse[(size_t)(w.b * p.Hq + w.h) * Sq + qrow] = (m[b] + fast_log2(lt)) * kLn2;
        }
        {
            char* obase = reinterpret_cast<char*>(p.o) + ((size_t)(w.b * p.Hq + w.h) * Sq) * RB;
#pragma unroll
            for (int i = 0; i < CPR; ++i) {
                const int cidx = lane + 64 * i;
                const int row = cidx / CPR, cc = cidx % CPR;
                const u32x4_t x = *reinterpret_cast<const u32x4_t*>(Os + row * RBP + cc * 16);
                if (q0w + row < Sq) *reinterpret_cast<u32x4_t*>(obase + (size_t)(q0w + row) * RB + cc * 16) = x;
            }
        }
        __syncthreads();  // slabs -> next part's ring
    }
}

template <class T, int D>
int launch_iw_t(const FwdArgs& a, hipStream_t stream, unsigned long long* dbg) {
    FwdIWParams p;
    p.q = a.q; p.k = a.k; p.v = a.v; p.o = a.o; p.lse = a.lse;
    p.B = a.B; p.Hq = a.Hq; p.Hkv = a.Hkv; p.Sq = a.Sq; p.Sk = a.Sk;
    p.c = a.scale * kLog2e;
    p.nqb = (a.Sq + kIWQBlock - 1) / kIWQBlock;
    p.pair = a.causal ? 1 : 0;
    p.nwork = p.pair ? (p.nqb + 1) / 2 : p.nqb;
    p.dbg = dbg;
    const dim3 grid((unsigned)(p.nwork * a.B * a.Hq)), block(256);
    const size_t lds = IWCfg<D>::LDS + 16;
    if (dbg != nullptr) {
        if constexpr (D == 128 && std::is_same<T, Bf16Traits>::value) {
            if (a.causal) {
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&fa_fwd_iw_kernel<T, D, true, true>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, lds);
                hipLaunchKernelGGL((fa_fwd_iw_kernel<T, D, true, true>), grid, block, lds, stream, p);
            } else {
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&fa_fwd_iw_kernel<T, D, false, true>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, lds);
                hipLaunchKernelGGL((fa_fwd_iw_kernel<T, D, false, true>), grid, block, lds, stream, p);
            }
            return (int)hipGetLastError();
        }
        return -1;
    }
    if (a.causal)
        hipLaunchKernelGGL((fa_fwd_iw_kernel<T, D, true>), grid, block, lds, stream, p);
    else
        hipLaunchKernelGGL((fa_fwd_iw_kernel<T, D, false>), grid, block, lds, stream, p);
    return (int)hipGetLastError();
}

template <class T, int D>
int set_attr_iw() {
    int rc = (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&fa_fwd_iw_kernel<T, D, true>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, IWCfg<D>::LDS + 16);
    rc |= (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&fa_fwd_iw_kernel<T, D, false>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, IWCfg<D>::LDS + 16);
    return rc;
}

}  // namespace

int launch_fwd_iw(const FwdArgs& a, hipStream_t stream) {
    if (a.dtype == kBF16) {
        if (a.D == 128) return launch_iw_t<Bf16Traits, 128>(a, stream, nullptr);
    }
    return -1;  // fp16 (P would leave the fp16 range without a running maximum) and D < 128: ping-pong kernel
}

int launch_fwd_iw_timeline(const FwdArgs& a, unsigned long long* dbg, hipStream_t stream) {
    if (a.dtype != kBF16 || a.D != 128) return -1;
    return launch_iw_t<Bf16Traits, 128>(a, stream, dbg);
}

int configure_fwd_iw() {
    return set_attr_iw<Bf16Traits, 128>();
}

}  // namespace aule_hip
