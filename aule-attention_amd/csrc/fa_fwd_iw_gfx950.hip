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
    float c;    // |scale| * log2(e)
    int negq;   // scale < 0: the sign goes into the Q fragments
    int nqb;    // 256-row Q blocks
    int nwork;  // work items per head: ceil(nqb/2) when pairing, else nqb
    int pair;   // process Q blocks (i, nqb-1-i) in one workgroup
    unsigned long long* dbg;  // timeline build only
};

constexpr int kIWQBlock = 256;
constexpr int kIWTile = 64;
constexpr int kIWTLMax = 512;

template <int D, int NB>
struct IWCfg {
    static constexpr int NT = 512 / NB;          // threads: NB = 2 -> 4 waves x 64 rows, NB = 1 -> 8 waves x 32 rows
    static constexpr int NWAVES = NT / 64;
    static constexpr int RB = D * 2;
    static constexpr int RBP = RB + 16;          // padded LDS row (K tile, epilogue slab)
    static constexpr int CPR = RB / 16;
    static constexpr int KTILE = kIWTile * RBP;
    static constexpr int VTILE = kIWTile * RB;   // [kv/4][d/16][4][16] sub-tiles
    static constexpr int NCHUNK = kIWTile * CPR;
    static constexpr int CH = NCHUNK / NT;       // 16-byte chunks per thread per tile
    static constexpr int KS = D / 16, DB = D / 32;
    static constexpr int OSLAB = 32 * NB * RBP;  // one wave's output rows (epilogue transpose)
    static constexpr int RING = 2 * KTILE + 2 * VTILE;
    static constexpr int LDS = RING > NWAVES * OSLAB ? RING : NWAVES * OSLAB;
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
template <bool AG>
__device__ __forceinline__ float acc_read(const float& a) {
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (AG) {
        float r;
        asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(r) : "a"(a));
        return r;
    } else {
        return a;
    }
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
template <bool AG>
__device__ __forceinline__ void scale_acc(f32x16_t& t, float alpha) {
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (!AG) {
#pragma unroll
        for (int r = 0; r < 16; ++r) t[r] *= alpha;
        return;
    }
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

// Fused per-slot softmax work of the fast pass, software-pipelined over four elements so that no instruction
// reads the result of the one or two before it (one wave per SIMD: nobody else fills a dependency bubble):
//     step k:  t_k = S_k (AGPR -> VGPR) ; x_{k-1} = t_{k-1}*c - m_ref ; p_{k-2} = exp2(x_{k-2}) ; l += p_{k-3}
//              and, when k-3 is odd, the bf16 pack of (p_{k-4}, p_{k-3}).
// One asm statement per step: hipcc pads separate asm statements with s_nop, and inline asm is invisible to
// its hazard recogniser (a transcendental's result must not be read by the next instruction: here it is
// first read one whole step later).
__device__ __forceinline__ void sp_step(const float& s_k, float& t_k, float t_km1, float& x_km1, float x_km2, float& p_km2,
                                        float p_km3, float& acc, float c, float nm) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("v_accvgpr_read_b32 %0, %4\n\tv_fma_f32 %1, %5, %8, %9\n\tv_exp_f32 %2, %6\n\tv_add_f32 %3, %3, %7"
                 : "=&v"(t_k), "=&v"(x_km1), "=&v"(p_km2), "+v"(acc)
                 : "a"(s_k), "v"(t_km1), "v"(x_km2), "v"(p_km3), "v"(c), "v"(nm));
#endif
}
__device__ __forceinline__ void sp_step_pk(const float& s_k, float& t_k, float t_km1, float& x_km1, float x_km2, float& p_km2,
                                           float p_km3, float p_km4, float& acc, float c, float nm, unsigned& packed) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("v_accvgpr_read_b32 %0, %5\n\tv_fma_f32 %1, %6, %10, %11\n\tv_exp_f32 %2, %7\n\tv_add_f32 %3, %3, %8\n\t"
                 "v_cvt_pk_bf16_f32 %4, %9, %8"
                 : "=&v"(t_k), "=&v"(x_km1), "=&v"(p_km2), "+v"(acc), "=&v"(packed)
                 : "a"(s_k), "v"(t_km1), "v"(x_km2), "v"(p_km3), "v"(p_km4), "v"(c), "v"(nm));
#endif
}
// VGPR-form kernels (8 waves x 32 rows, at most 256 registers: the MFMA results are in arch VGPRs): the same step
// without the AGPR read -- x_{k-1} = S_{k-1}*c - m_ref ; p_{k-2} = exp2(x_{k-2}) ; l += p_{k-3} [; pack (p_{k-4}, p_{k-3})]
__device__ __forceinline__ void sp_step_v(float s_km1, float& x_km1, float x_km2, float& p_km2, float p_km3, float& acc, float c, float nm) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("v_fma_f32 %0, %3, %6, %7\n\tv_exp_f32 %1, %4\n\tv_add_f32 %2, %2, %5"
                 : "=&v"(x_km1), "=&v"(p_km2), "+v"(acc) : "v"(s_km1), "v"(x_km2), "v"(p_km3), "v"(c), "v"(nm));
#endif
}
__device__ __forceinline__ void sp_step_pk_v(float s_km1, float& x_km1, float x_km2, float& p_km2, float p_km3, float p_km4, float& acc,
                                             float c, float nm, unsigned& packed) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("v_fma_f32 %0, %4, %8, %9\n\tv_exp_f32 %1, %5\n\tv_add_f32 %2, %2, %6\n\tv_cvt_pk_bf16_f32 %3, %7, %6"
                 : "=&v"(x_km1), "=&v"(p_km2), "+v"(acc), "=&v"(packed)
                 : "v"(s_km1), "v"(x_km2), "v"(p_km3), "v"(p_km4), "v"(c), "v"(nm));
#endif
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

// FAST pass: the row maximum of the first tile stays the softmax reference of the row (exact algebra, see
// fa_fwd_pp_gfx950.hip "fixed-reference softmax"); valid while every row sum stays in [2^-100, 2^110].
// SAFE pass: classic online softmax, tile by tile, not software-pipelined; a workgroup re-runs a Q block in
// this mode when the fast pass left the range.
template <class T, int D, bool CAUSAL, int NB = 2, bool TL = false>
__global__ void __launch_bounds__(512 / NB) fa_fwd_iw_kernel(const FwdIWParams p) {
    using C = IWCfg<D, NB>;
    constexpr int NT = C::NT;
    constexpr bool AG = (NB == 2);   // 512-register budget: AGPR-form MFMAs (S and O live in AGPRs)
    constexpr int NE = 16 * NB;      // softmax elements per half (kv half h of all blocks of the wave)
    using v8 = typename T::v8;
    static_assert(std::is_same<T, Bf16Traits>::value, "the fast pass relies on bf16's fp32 exponent range for P");
    constexpr int RB = C::RB, RBP = C::RBP, CPR = C::CPR, KTILE = C::KTILE, VTILE = C::VTILE;
    constexpr int CH = C::CH, KS = C::KS, DB = C::DB;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const Ks = smem;
    char* const Vs = smem + 2 * KTILE;
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
    const float c = p.c;  // |scale| * log2(e); the sign goes into Q

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
        const int cidx = tid + NT * i;
        const int row = cidx / CPR, cc = cidx % CPR;
        k_g[i] = row * RB + cc * 16;
        k_lds[i] = row * RBP + cc * 16;
        const int bidx = (tid >> 3) + (NT / 8) * i;  // sub-tile index = kv4 * (D/16) + d16
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
        for (int i = 0; i < CH; ++i) *reinterpret_cast<u32x4_t*>(Vs + buf * VTILE + tid * 16 + i * (NT * 16)) = vst[i];
    };

    if (tid == 0) *flag = 0;

    const int nparts = (p.pair && (p.nqb - 1 - w.blk) != w.blk) ? 2 : 1;
    for (int part = 0; part < nparts; ++part) {
        const int qb = p.pair ? (part == 0 ? p.nqb - 1 - w.blk : w.blk) : w.blk;
        const int q0w = qb * kIWQBlock + wave * (32 * NB);

        const int kv_hi = CAUSAL ? min(Sk, qb * kIWQBlock + kIWQBlock) : Sk;
        const int nt = (kv_hi + kIWTile - 1) / kIWTile;          // tiles staged by the workgroup (>= 1)
        const int wave_kv_hi = CAUSAL ? min(Sk, q0w + 32 * NB) : Sk;  // keys visible to this wave
        const int na = (wave_kv_hi + kIWTile - 1) / kIWTile;     // tiles this wave computes (a prefix, >= 1)

        // Q fragments (B operand of S^T = K.Q^T): lane (q, hi) of block b holds d = 16ks+8hi..+7
        v8 qf[NB][KS];
        {
            const size_t qhead = (size_t)(w.b * p.Hq + w.h) * Sq * RB;
            const __amdgpu_buffer_rsrc_t qrs = iw_srd(reinterpret_cast<const char*>(p.q) + qhead, (unsigned)Sq * RB);
            const unsigned flip = p.negq ? 0x80008000u : 0u;
            u32x4_t qx[NB][KS];
#pragma unroll
            for (int b = 0; b < NB; ++b)
#pragma unroll
                for (int ks = 0; ks < KS; ++ks)
                    qx[b][ks] = __builtin_amdgcn_raw_buffer_load_b128(qrs, (q0w + 32 * b + l31) * RB + (2 * ks + hi) * 16, 0, 0);
#pragma unroll
            for (int b = 0; b < NB; ++b)
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    u32x4_t x = qx[b][ks];
                    x[0] ^= flip; x[1] ^= flip; x[2] ^= flip; x[3] ^= flip;
                    qf[b][ks] = as_v8<T>(x);
                }
        }

        f32x16_t o[NB][DB];
        float m[NB], l[NB];
        f32x16_t s[2][NB][2];   // [tile parity][block][kv half]
        v8 pb[2][NB][2][2];     // [tile parity][block][kv half][k-step inside the half]

        auto zero_state = [&](float m0) __attribute__((always_inline)) {
#pragma unroll
            for (int b = 0; b < NB; ++b) {
#pragma unroll
                for (int d = 0; d < DB; ++d)
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[b][d][r] = 0.f;
                m[b] = m0;
                l[b] = 0.f;
            }
        };
        f32x16_t z;
#pragma unroll
        for (int r = 0; r < 16; ++r) z[r] = 0.f;

        // masked / unmasked x of one S element of tile `kv0` (log2 units, relative to nm = -m_ref)
        auto xmask = [&](float x, int kv0, int blk, int h, int r, auto sm_tag) __attribute__((always_inline)) -> float {
            if constexpr (decltype(sm_tag)::value == 2) {
                const int kv = kv0 + 32 * h + crow(r, hi);
                const bool vis = (kv < Sk) && (!CAUSAL || kv <= q0w + 32 * blk + l31);
                x = vis ? x : -INFINITY;
            }
            return x;
        };

        // ---- softmax of one half (kv half H of both blocks, 32 elements) of the tile with parity PAR, element e
        //      handled in MFMA slot e of the caller (or back to back when there is no MFMA stream).
        //      Element order = consumption order of the PV phase: (block 0, r 0..7), (block 1, r 0..7),
        //      (block 0, r 8..15), (block 1, r 8..15).
        auto elem_blk = [](int e) { return NB == 2 ? ((e >> 3) & 1) : 0; };
        auto elem_r = [](int e) { return NB == 2 ? ((e & 7) + 8 * (e >> 4)) : e; };

        // ---- phase A of tile j (parity PAR): S_{j+1} = K_{j+1} Q^T for both blocks (HAS_QK), with the second
        //      half of softmax(S_j) in its slots (SM: 0 none, 1 plain, 2 masked).
        // ---- phase B of tile j: O += V_j^T P_j for both blocks, with the first half of softmax(S_{j+1}).
        // One pipeline step k (0 .. 34) of a half: see sp_step.  Boundary steps and masked tiles go through the
        // single pinned instructions.
        struct Pipe { float t[NE + 4], x[NE + 4], p[NE + 4]; };
        auto half_softmax_step = [&](auto par_tag, auto h_tag, auto sm_tag, int kv0, int k, Pipe& q, const float (&nm)[NB]) __attribute__((always_inline)) {
            constexpr int PAR = decltype(par_tag)::value, H = decltype(h_tag)::value, SM = decltype(sm_tag)::value;
            auto put_pack = [&](int e_hi, unsigned pk) __attribute__((always_inline)) {  // pack of elements (e_hi - 1, e_hi)
                const int b2 = elem_blk(e_hi), i2 = elem_r(e_hi) >> 1;
                u32x4_t t = __builtin_bit_cast(u32x4_t, pb[PAR][b2][H][i2 >> 2]);
                t[i2 & 3] = pk;
                pb[PAR][b2][H][i2 >> 2] = as_v8<T>(t);
            };
            if constexpr (SM == 1) {
                if (k >= 4 && k <= NE - 1) {
                    const float sk_ = s[PAR][elem_blk(k)][H][elem_r(k)];
                    if constexpr (AG) {
                        if ((k - 3) & 1) {
                            unsigned pk;
                            sp_step_pk(sk_, q.t[k], q.t[k - 1], q.x[k - 1], q.x[k - 2], q.p[k - 2], q.p[k - 3], q.p[k - 4],
                                       l[elem_blk(k - 3)], c, nm[elem_blk(k - 1)], pk);
                            put_pack(k - 3, pk);
                        } else {
                            sp_step(sk_, q.t[k], q.t[k - 1], q.x[k - 1], q.x[k - 2], q.p[k - 2], q.p[k - 3],
                                    l[elem_blk(k - 3)], c, nm[elem_blk(k - 1)]);
                        }
                    } else {
                        q.t[k] = sk_;  // already in an arch VGPR
                        if ((k - 3) & 1) {
                            unsigned pk;
                            sp_step_pk_v(q.t[k - 1], q.x[k - 1], q.x[k - 2], q.p[k - 2], q.p[k - 3], q.p[k - 4],
                                         l[elem_blk(k - 3)], c, nm[elem_blk(k - 1)], pk);
                            put_pack(k - 3, pk);
                        } else {
                            sp_step_v(q.t[k - 1], q.x[k - 1], q.x[k - 2], q.p[k - 2], q.p[k - 3],
                                      l[elem_blk(k - 3)], c, nm[elem_blk(k - 1)]);
                        }
                    }
                    return;
                }
            }
            if (k <= NE - 1) {
                const float sk_ = s[PAR][elem_blk(k)][H][elem_r(k)];
                q.t[k] = acc_read<AG>(sk_);
            }
            if (k >= 1 && k - 1 <= NE - 1) {
                const int e = k - 1;
                float x = fma_pinned(q.t[e], c, nm[elem_blk(e)]);
                x = xmask(x, kv0, elem_blk(e), H, elem_r(e), sm_tag);
                q.x[e] = x;
            }
            if (k >= 2 && k - 2 <= NE - 1) q.p[k - 2] = exp2_pinned(q.x[k - 2]);
            if (k >= 3 && k - 3 <= NE - 1) {
                const int e = k - 3;
                add_pinned(l[elem_blk(e)], q.p[e]);
                if (e & 1) put_pack(e, pack_bf16_pinned(q.p[e - 1], q.p[e]));
            }
        };
        constexpr int kSteps = NE + 3;

        auto phaseA = [&](auto par_tag, auto qk_tag, auto sm_tag, int kv0, const float (&nm)[NB]) __attribute__((always_inline)) {
            constexpr int PAR = decltype(par_tag)::value, NXT = PAR ^ 1;
            constexpr bool HAS_QK = decltype(qk_tag)::value != 0;
            constexpr int SM = decltype(sm_tag)::value;
            constexpr int NOP = 2 * KS;            // K operands: t = 2 ks + h, each feeds both blocks
            constexpr int kAhead = 2;
            Pipe pq;
            if constexpr (HAS_QK) {
                const char* kb = Ks + NXT * KTILE + ka_base;
                u32x4_t kf[NOP];
                auto rd = [&](int t) __attribute__((always_inline)) {
                    kf[t] = *reinterpret_cast<const u32x4_t*>(kb + (t >> 1) * 32 + (t & 1) * 32 * RBP);
                };
#pragma unroll
                for (int t = 0; t < kAhead && t < NOP; ++t) rd(t);
#pragma unroll
                for (int sl = 0; sl < NB * NOP; ++sl) {
                    const int t = sl / NB, b = sl % NB, ks = t >> 1, h = t & 1;
                    if (b == 0 && t + kAhead < NOP) rd(t + kAhead);
                    s[NXT][b][h] = T::mfma(as_v8<T>(kf[t]), qf[b][ks], ks == 0 ? z : s[NXT][b][h]);
                    if constexpr (SM != 0) {
#pragma unroll
                        for (int k = (kSteps * sl) / (NB * NOP); k < (kSteps * (sl + 1)) / (NB * NOP); ++k)
                            half_softmax_step(par_tag, ic<1>{}, sm_tag, kv0, k, pq, nm);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else if constexpr (SM != 0) {
#pragma unroll
                for (int k = 0; k < kSteps; ++k) half_softmax_step(par_tag, ic<1>{}, sm_tag, kv0, k, pq, nm);
            }
        };
        auto phaseB = [&](auto par_tag, auto pv_tag, auto sm_tag, int kv0_next, const float (&nm)[NB]) __attribute__((always_inline)) {
            constexpr int PAR = decltype(par_tag)::value, NXT = PAR ^ 1;
            constexpr bool HAS_PV = decltype(pv_tag)::value != 0;
            constexpr int SM = decltype(sm_tag)::value;
            constexpr int NOP = 4 * DB;            // V operands: t = sk * DB + d, each feeds both blocks
            constexpr int kAhead = 2;
            constexpr int kLead = 3;               // S_{j+1} was written by the last MFMAs of phase A
            Pipe pq;
            if constexpr (HAS_PV) {
                const char* vb = Vs + PAR * VTILE + va_off;
                s16x4_t a0[NOP], a1[NOP];
                auto rd = [&](int t) __attribute__((always_inline)) {
                    const int sk = t / DB, d = t % DB;
                    const int off = ((4 * sk) * (D / 16) + 2 * d) * 128;
                    a0[t] = lds_tr16(vb + off);
                    a1[t] = lds_tr16(vb + off + 2 * (D / 16) * 128);
                };
                constexpr int NSL = NB * NOP;
                auto first_step = [&](int sl) __attribute__((always_inline)) { return sl <= kLead ? 0 : (kSteps * (sl - kLead)) / (NSL - kLead); };
#pragma unroll
                for (int t = 0; t < kAhead && t < NOP; ++t) rd(t);
#pragma unroll
                for (int sl = 0; sl < NSL; ++sl) {
                    const int t = sl / NB, b = sl % NB, sk = t / DB, d = t % DB;
                    if (b == 0 && t + kAhead < NOP) rd(t + kAhead);
                    o[b][d] = T::mfma(as_v8<T>(a0[t], a1[t]), pb[PAR][b][sk >> 1][sk & 1], o[b][d]);
                    if constexpr (SM != 0) {
#pragma unroll
                        for (int k = first_step(sl); k < first_step(sl + 1); ++k)
                            half_softmax_step(ic<NXT>{}, ic<0>{}, sm_tag, kv0_next, k, pq, nm);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else if constexpr (SM != 0) {
#pragma unroll
                for (int k = 0; k < kSteps; ++k) half_softmax_step(ic<NXT>{}, ic<0>{}, sm_tag, kv0_next, k, pq, nm);
            }
        };

        // =========================== FAST pass ===========================
        auto run_fast = [&]() __attribute__((always_inline)) {
            zero_state(0.f);
            issue_k(0);
            issue_v(0);
            write_k(0);
            write_v(0);
            issue_k(kIWTile);
            write_k(1);
            __syncthreads();
            float nm[NB];
#pragma unroll
            for (int b = 0; b < NB; ++b) nm[b] = 0.f;
            // pre-phase: S_0 (parity 0), its row maximum = the reference, first half of softmax(S_0)
            phaseA(ic<1>{}, ic<1>{}, ic<0>{}, 0, nm);  // "tile -1" has parity 1: writes s[0] from K buffer 0
            {
                const bool last0 = (na == 1);
#pragma unroll
                for (int b = 0; b < NB; ++b) {
                    float mx = -INFINITY;
#pragma unroll
                    for (int h = 0; h < 2; ++h)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            float x = acc_read<AG>(s[0][b][h][r]);
                            if (last0) x = xmask(x, 0, b, h, r, ic<2>{});
                            mx = fmaxf(mx, x);
                        }
                    mx = fmaxf(mx, xhalf(mx));
                    m[b] = (mx == -INFINITY) ? 0.f : mx * c;
                    nm[b] = -m[b];
                }
                if (last0) phaseB(ic<1>{}, ic<0>{}, ic<2>{}, 0, nm);
                else phaseB(ic<1>{}, ic<0>{}, ic<1>{}, 0, nm);
            }
            __syncthreads();  // nobody may still read K_0 when K_2 is staged into its buffer

            // One tile step (parity PAR = j & 1).  KIND: 0 steady (tile j+1 is not the last), 1 tile j+1 is the
            // last one (its softmax is masked), 2 tile j is the last, 3 idle (staging and barrier only).
            auto tile_step = [&](int j, auto par_tag, auto kind_tag) __attribute__((always_inline)) {
                constexpr int PAR = decltype(par_tag)::value, NXT = PAR ^ 1, KIND = decltype(kind_tag)::value;
                stamp();
                issue_v((j + 1) * kIWTile);
                issue_k((j + 2) * kIWTile);
                const int kv0 = j * kIWTile;
                if constexpr (KIND == 0) {
                    phaseA(par_tag, ic<1>{}, ic<1>{}, kv0, nm);
                    stamp();
                    phaseB(par_tag, ic<1>{}, ic<1>{}, kv0 + kIWTile, nm);
                } else if constexpr (KIND == 1) {
                    phaseA(par_tag, ic<1>{}, ic<1>{}, kv0, nm);
                    stamp();
                    phaseB(par_tag, ic<1>{}, ic<2>{}, kv0 + kIWTile, nm);
                } else if constexpr (KIND == 2) {
                    phaseA(par_tag, ic<0>{}, ic<2>{}, kv0, nm);
                    stamp();
                    phaseB(par_tag, ic<1>{}, ic<0>{}, kv0 + kIWTile, nm);
                } else {
                    stamp();
                }
                stamp();
                __builtin_amdgcn_sched_barrier(0);
                write_v(NXT);
                write_k(PAR);
                __syncthreads();
                __builtin_amdgcn_sched_barrier(0);
                stamp();
            };
            int j = 0;
            for (; j + 3 < na; j += 2) {
                tile_step(j, ic<0>{}, ic<0>{});
                tile_step(j + 1, ic<1>{}, ic<0>{});
            }
            const int rem = na - j;  // 1, 2 or 3; j is even
            if (rem == 3) {
                tile_step(j, ic<0>{}, ic<0>{});
                tile_step(j + 1, ic<1>{}, ic<1>{});
                tile_step(j + 2, ic<0>{}, ic<2>{});
            } else if (rem == 2) {
                tile_step(j, ic<0>{}, ic<1>{});
                tile_step(j + 1, ic<1>{}, ic<2>{});
            } else {
                tile_step(j, ic<0>{}, ic<2>{});
            }
            for (j = na; j < nt; ++j) {
                if (j & 1) tile_step(j, ic<1>{}, ic<3>{});
                else tile_step(j, ic<0>{}, ic<3>{});
            }
        };

        // =========================== SAFE pass ===========================
        auto run_safe = [&]() __attribute__((always_inline)) {
            zero_state(-INFINITY);
            float nm0[NB];
#pragma unroll
            for (int b = 0; b < NB; ++b) nm0[b] = 0.f;
            for (int j = 0; j < nt; ++j) {
                __syncthreads();  // previous tile's readers are done with buffer 1 / 0
                issue_k(j * kIWTile);
                issue_v(j * kIWTile);
                write_k(1);
                write_v(0);
                __syncthreads();
                if (j < na) {
                    phaseA(ic<0>{}, ic<1>{}, ic<0>{}, 0, nm0);  // s[1] = K(buffer 1) Q^T
                    const bool need_mask = (CAUSAL && (j * kIWTile + kIWTile - 1 > q0w)) || (j * kIWTile + kIWTile > Sk);
#pragma unroll
                    for (int b = 0; b < NB; ++b) {
                        float x[32];
                        float mx = -INFINITY;
#pragma unroll
                        for (int e = 0; e < 32; ++e) {
                            const int h = e >> 4, r = e & 15;
                            x[e] = acc_read<AG>(s[1][b][h][r]) * c;
                            if (need_mask) x[e] = xmask(x[e], j * kIWTile, b, h, r, ic<2>{});
                            mx = fmaxf(mx, x[e]);
                        }
                        mx = fmaxf(mx, xhalf(mx));
                        const float m_new = fmaxf(m[b], mx);
                        const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
                        const float alpha = fast_exp2(m[b] - m_use);  // exp2(-inf) = 0 on the first tile (O = l = 0)
                        m[b] = m_new;
                        l[b] *= alpha;
                        if (__builtin_amdgcn_ballot_w64(alpha != 1.0f) != 0) {
#pragma unroll
                            for (int d = 0; d < DB; ++d) scale_acc<AG>(o[b][d], alpha);
                        }
                        float a2 = 0.f;
#pragma unroll
                        for (int e = 0; e < 32; e += 2) {
                            const float p0 = fast_exp2(x[e] - m_use), p1 = fast_exp2(x[e + 1] - m_use);
                            a2 += p0 + p1;
                            const int h = e >> 4, i = (e & 15) >> 1;
                            u32x4_t t = __builtin_bit_cast(u32x4_t, pb[0][b][h][i >> 2]);
                            t[i & 3] = T::pack2(p0, p1);
                            pb[0][b][h][i >> 2] = as_v8<T>(t);
                        }
                        l[b] += a2;
                    }
                    phaseB(ic<0>{}, ic<1>{}, ic<0>{}, 0, nm0);  // O += V(buffer 0)^T P (pb[0])
                }
            }
            __syncthreads();
        };

        run_fast();
        // range check of the fast pass (NaN fails it too); one verdict per workgroup
        {
            bool ok = true;
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                const float lt = l[b] + xhalf(l[b]);
                ok = ok && (lt > 0x1p-100f) && (lt < 0x1p110f);
            }
            if (__builtin_amdgcn_ballot_w64(!ok) != 0 && lane == 0) *flag = 1;
            __syncthreads();
            const int redo = *flag;
            __syncthreads();
            if (redo) {
                if (tid == 0) *flag = 0;
                run_safe();
            }
        }

        // ---- epilogue: O = O^T / l through a per-wave LDS slab (whole 16-byte row chunks to global);
        //      LSE = (m + log2 l) * ln2
        char* const Os = smem + wave * C::OSLAB;
#pragma unroll
        for (int b = 0; b < NB; ++b) {
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
            for (int i = 0; i < (32 * NB * CPR) / 64; ++i) {
                const int cidx = lane + 64 * i;
                const int row = cidx / CPR, cc = cidx % CPR;
                const u32x4_t x = *reinterpret_cast<const u32x4_t*>(Os + row * RBP + cc * 16);
                if (q0w + row < Sq) *reinterpret_cast<u32x4_t*>(obase + (size_t)(q0w + row) * RB + cc * 16) = x;
            }
        }
        __syncthreads();  // slabs -> next part's ring
    }
}

template <class T, int D, int NB>
int launch_iw_t(const FwdArgs& a, hipStream_t stream, unsigned long long* dbg) {
    FwdIWParams p;
    p.q = a.q; p.k = a.k; p.v = a.v; p.o = a.o; p.lse = a.lse;
    p.B = a.B; p.Hq = a.Hq; p.Hkv = a.Hkv; p.Sq = a.Sq; p.Sk = a.Sk;
    p.c = a.scale * kLog2e;
    p.negq = p.c < 0.f;
    p.c = p.negq ? -p.c : p.c;
    if (p.c == 0.f) p.c = 1e-30f;
    p.nqb = (a.Sq + kIWQBlock - 1) / kIWQBlock;
    p.pair = a.causal ? 1 : 0;
    p.nwork = p.pair ? (p.nqb + 1) / 2 : p.nqb;
    p.dbg = dbg;
    const dim3 grid((unsigned)(p.nwork * a.B * a.Hq)), block(IWCfg<D, NB>::NT);
    const size_t lds = IWCfg<D, NB>::LDS + 16;
    auto go = [&](auto kern) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        hipLaunchKernelGGL(kern, grid, block, lds, stream, p);
    };
    if (dbg != nullptr) {
        if (a.causal) go(&fa_fwd_iw_kernel<T, D, true, NB, true>);
        else go(&fa_fwd_iw_kernel<T, D, false, NB, true>);
    } else {
        if (a.causal) go(&fa_fwd_iw_kernel<T, D, true, NB, false>);
        else go(&fa_fwd_iw_kernel<T, D, false, NB, false>);
    }
    return (int)hipGetLastError();
}

// AULE_HIP_FWD_KERNEL = "iw" (4 waves x 64 rows, AGPR form) | "iw1" (8 waves x 32 rows, VGPR form)
static int iw_blocks() {
    static const int v = [] {
        const char* e = getenv("AULE_HIP_FWD_KERNEL");
        return (e != nullptr && e[0] == 'i' && e[1] == 'w' && e[2] == '1') ? 1 : 2;
    }();
    return v;
}

}  // namespace

int launch_fwd_iw(const FwdArgs& a, hipStream_t stream) {
    if (a.dtype == kBF16 && a.D == 128)
        return iw_blocks() == 1 ? launch_iw_t<Bf16Traits, 128, 1>(a, stream, nullptr) : launch_iw_t<Bf16Traits, 128, 2>(a, stream, nullptr);
    return -1;  // fp16 (P would leave the fp16 range without a running maximum) and D < 128: ping-pong kernel
}

int launch_fwd_iw_timeline(const FwdArgs& a, unsigned long long* dbg, hipStream_t stream) {
    if (a.dtype != kBF16 || a.D != 128) return -1;
    return iw_blocks() == 1 ? launch_iw_t<Bf16Traits, 128, 1>(a, stream, dbg) : launch_iw_t<Bf16Traits, 128, 2>(a, stream, dbg);
}

int configure_fwd_iw() { return 0; }  // the launcher sets the LDS attribute itself

}  // namespace aule_hip
