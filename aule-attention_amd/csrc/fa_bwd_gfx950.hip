// fa_bwd_gfx950.hip -- FlashAttention-2 backward for MI355X (gfx950 / CDNA4).
//
// Replaces, behind aule_attention_backward_ex, the reference's Triton backward
// (python/aule/triton_flash_amd.py:447-500 + kernels :247-380; generic twin
// python/aule/triton_flash.py:478-526, :242-379) and the Vulkan BackwardPipeline
// (src/attention_backward_pipeline.zig:472-537, shaders/attention_backward_f32.comp).
// Math: delta_i = sum_d O_id dO_id ; p_ij = exp(s_ij - LSE_i)
// (0 where masked) ; dV += P^T dO ; dP = dO V^T ; dS = P o (dP - delta) * scale ;
// dQ = dS K ; dK = dS^T Q ; GQA: dK/dV reduce over the query heads of the group.
//
// Structure -- three launches, no atomics, deterministic:
//   1. fa_bwd_delta   : delta = rowsum(O o dO)  (HBM-bound stream)
//   2. fa_bwd_dq      : one workgroup per 256-row Q block (8 waves x 32 rows) loops
//                       over 64-row KV tiles; lane owns one query row, so LSE/delta
//                       are lane-local scalars (same swapped layout as the forward).
//   3. fa_bwd_dkdv    : one workgroup per 128-row KV block (4 waves x 32 rows, one
//                       wave per SIMD, K/V fragments resident in registers) loops
//                       over the group's query heads and 32-row Q tiles; lane owns
//                       one key row, dK^T / dV^T accumulate in registers; the GQA
//                       group reduction is the loop, not an atomic.
// The softmax is recomputed twice (once per kernel): 7 tile matmuls instead of the
// 5 of the atomic formulation, in exchange for no fp32 atomics on dQ.
#include "fa_device.h"
#include "fa_kernels.h"

namespace aule_hip {
namespace {

// ------------------------------------------------------------------ delta ----
struct DeltaParams {
    const void* o;
    const void* dout;
    float* delta;
    long long rows;  // B*Hq*Sq
};

// CPR 16-byte chunks per row; one lane per chunk, CPR-lane groups reduce by shuffle.
template <class T, int D>
__global__ void __launch_bounds__(256) fa_bwd_delta_kernel(const DeltaParams p) {
    constexpr int CPR = D * 2 / 16;
    constexpr int RPB = 256 / CPR;  // rows per block
    const int tid = threadIdx.x;
    const int sub = tid % CPR;
    const long long row = (long long)blockIdx.x * RPB + tid / CPR;
    float acc = 0.f;
    if (row < p.rows) {
        const u32x4_t a = reinterpret_cast<const u32x4_t*>(p.o)[row * CPR + sub];
        const u32x4_t b = reinterpret_cast<const u32x4_t*>(p.dout)[row * CPR + sub];
#pragma unroll
        for (int j = 0; j < 4; ++j) acc += T::lo(a[j]) * T::lo(b[j]) + T::hi(a[j]) * T::hi(b[j]);
    }
#pragma unroll
    for (int off = CPR / 2; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if (row < p.rows && sub == 0) p.delta[row] = acc;
}

template <int D>
__global__ void __launch_bounds__(256) fa_bwd_delta_f32_kernel(const DeltaParams p) {
    constexpr int CPR = D * 4 / 16;  // 8, 16 or 32 lanes per row
    constexpr int RPB = 256 / CPR;
    const int tid = threadIdx.x;
    const int sub = tid % CPR;
    const long long row = (long long)blockIdx.x * RPB + tid / CPR;
    float acc = 0.f;
    if (row < p.rows) {
        const f32x4_t a = reinterpret_cast<const f32x4_t*>(p.o)[row * CPR + sub];
        const f32x4_t b = reinterpret_cast<const f32x4_t*>(p.dout)[row * CPR + sub];
        acc = a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3];
    }
#pragma unroll
    for (int off = CPR / 2; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if (row < p.rows && sub == 0) p.delta[row] = acc;
}

// ------------------------------------------------------------- shared bits ----
struct BwdParams {
    const void* q;
    const void* k;
    const void* v;
    const void* dout;
    const float* lse;
    const float* delta;
    void* dq;
    void* dk;
    void* dv;
    int B, Hq, Hkv, Sq, Sk;
    float c;      // scale * log2(e)   (sign kept; no max is taken in the backward)
    float scale;  // applied to dQ / dK in the epilogue
    int nblk;     // Q blocks (dq kernel) or KV blocks (dkdv kernel)
};

// Row-major image swizzle (shared with the forward's K image).
template <int D>
__device__ __forceinline__ int rswz(int row) {
    if constexpr (D == 128) return row & 15;
    else if constexpr (D == 64) return (row >> 1) & 7;
    else return (row >> 2) & 3;
}
// byte offset of 16-byte chunk cc of row `row` in a row-major swizzled [rows][D] 16-bit image
template <int D>
__device__ __forceinline__ int rm_off(int row, int cc) {
    return row * (D * 2) + ((cc ^ rswz<D>(row)) << 4);
}
// byte offset of the same chunk in the [row/4][D/16][4][16] sub-tiled image (transpose-read source)
template <int D>
__device__ __forceinline__ int st_off(int row, int cc) {
    return ((row >> 2) * (D / 16) + (cc >> 1)) * 128 + (row & 3) * 32 + (cc & 1) * 16;
}

// Write the lane's 4 contiguous-d results of 4 accumulator registers as one 8-byte store
template <class T>
__device__ __forceinline__ void store4(char* row_ptr, int d_elem, float a, float b, float c2, float d2) {
    u32x2_t u;
    u[0] = T::pack2(a, b);
    u[1] = T::pack2(c2, d2);
    *reinterpret_cast<u32x2_t*>(row_ptr + d_elem * 2) = u;
}

// --------------------------------------------------------------- dQ kernel ----
constexpr int kDqQBlock = 256;
constexpr int kDqKV = 64;

template <int D>
struct DqCfg {
    static constexpr int RB = D * 2, CPR = RB / 16, TILE = kDqKV * RB, NCHUNK = TILE / 16;
    static constexpr int CH = (NCHUNK + 511) / 512, KS = D / 16, DB = D / 32;
    static constexpr int STAGE = 3 * TILE;  // K row-major, K sub-tiled, V row-major
    static constexpr int LDS = 2 * STAGE;
};

template <class T, int D, bool CAUSAL>
__global__ void __launch_bounds__(512) fa_bwd_dq_kernel(const BwdParams p) {
    using Cfg = DqCfg<D>;
    using v8 = typename T::v8;
    constexpr int RB = Cfg::RB, CPR = Cfg::CPR, TILE = Cfg::TILE, CH = Cfg::CH, KS = Cfg::KS, DB = Cfg::DB;
    constexpr int STAGE = Cfg::STAGE;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const WorkItem w = decode_work(blockIdx.x, p.B, p.Hq, p.Hkv, p.nblk, CAUSAL);
    const int Sq = p.Sq, Sk = p.Sk;
    const int q0w = w.blk * kDqQBlock + wave * 32;
    const int qrow = q0w + l31;
    const int qr = qrow < Sq ? qrow : Sq - 1;
    const size_t qbase = (size_t)(w.b * p.Hq + w.h) * Sq;

    const u32x4_t* __restrict__ kg = reinterpret_cast<const u32x4_t*>(p.k) + (size_t)(w.b * p.Hkv + w.hk) * Sk * CPR;
    const u32x4_t* __restrict__ vg = reinterpret_cast<const u32x4_t*>(p.v) + (size_t)(w.b * p.Hkv + w.hk) * Sk * CPR;

    v8 qf[KS], dof[KS];
    {
        const u32x4_t* qp = reinterpret_cast<const u32x4_t*>(p.q) + (qbase + qr) * CPR;
        const u32x4_t* gp = reinterpret_cast<const u32x4_t*>(p.dout) + (qbase + qr) * CPR;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            qf[ks] = as_v8<T>(qp[2 * ks + hi]);
            dof[ks] = as_v8<T>(gp[2 * ks + hi]);
        }
    }
    const float lse2 = p.lse[qbase + qr] * kLog2e;
    const float delta = p.delta[qbase + qr];
    const float c = p.c;

    int st_row[CH], st_cc[CH];
#pragma unroll
    for (int i = 0; i < CH; ++i) {
        const int cidx = tid + 512 * i;
        st_row[i] = cidx / CPR;
        st_cc[i] = cidx % CPR;
    }
    int a_off[KS];  // row-major A-operand offsets (row l31, chunk 2ks+hi)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) a_off[ks] = rm_off<D>(l31, 2 * ks + hi);
    const int tr_off = hi * (D / 16) * 128 + ((lane >> 4) & 1) * 128 + (lane & 15) * 8;

    const int kv_hi = CAUSAL ? min(Sk, w.blk * kDqQBlock + kDqQBlock) : Sk;
    const int nt = (kv_hi + kDqKV - 1) / kDqKV;
    const int wave_kv_hi = CAUSAL ? min(Sk, q0w + 32) : Sk;

    u32x4_t kst[CH], vst[CH];
    auto issue_loads = [&](int kv0) {
#pragma unroll
        for (int i = 0; i < CH; ++i)
            if (Cfg::NCHUNK % 512 == 0 || tid + 512 * i < Cfg::NCHUNK) {
                int r = kv0 + st_row[i];
                r = r < Sk ? r : Sk - 1;
                kst[i] = kg[(size_t)r * CPR + st_cc[i]];
                vst[i] = vg[(size_t)r * CPR + st_cc[i]];
            }
    };
    auto write_stage = [&](int buf) {
        char* base = smem + buf * STAGE;
#pragma unroll
        for (int i = 0; i < CH; ++i)
            if (Cfg::NCHUNK % 512 == 0 || tid + 512 * i < Cfg::NCHUNK) {
                *reinterpret_cast<u32x4_t*>(base + rm_off<D>(st_row[i], st_cc[i])) = kst[i];
                *reinterpret_cast<u32x4_t*>(base + TILE + st_off<D>(st_row[i], st_cc[i])) = kst[i];
                *reinterpret_cast<u32x4_t*>(base + 2 * TILE + rm_off<D>(st_row[i], st_cc[i])) = vst[i];
            }
    };

    f32x16_t acc[DB];
#pragma unroll
    for (int d = 0; d < DB; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[d][r] = 0.f;

    issue_loads(0);
    write_stage(0);
    __syncthreads();

    for (int t = 0; t < nt; ++t) {
        const int cur = t & 1;
        const int kv0 = t * kDqKV;
        if (t + 1 < nt) issue_loads(kv0 + kDqKV);
        if (kv0 < wave_kv_hi) {
            const char* krm = smem + cur * STAGE;
            const char* kst_img = krm + TILE + tr_off;
            const char* vrm = krm + 2 * TILE;
#pragma unroll
            for (int sb = 0; sb < 2; ++sb) {
                if (sb == 1 && kv0 + 32 >= wave_kv_hi) break;  // second half fully masked (wave-uniform)
                f32x16_t s, dp;
#pragma unroll
                for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const u32x4_t ka = *reinterpret_cast<const u32x4_t*>(krm + a_off[ks] + sb * 32 * RB);
                    const u32x4_t va = *reinterpret_cast<const u32x4_t*>(vrm + a_off[ks] + sb * 32 * RB);
                    s = T::mfma(as_v8<T>(ka), qf[ks], s);      // S^T  = K  . Q^T
                    dp = T::mfma(as_v8<T>(va), dof[ks], dp);   // dP^T = V  . dO^T
                }
                const bool need_mask = (CAUSAL && (kv0 + sb * 32 + 31 > q0w)) || (kv0 + sb * 32 + 32 > Sk);
                float ds[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float pv = fast_exp2(__builtin_fmaf(s[r], c, -lse2));
                    if (need_mask) {
                        const int kv = kv0 + sb * 32 + crow(r, hi);
                        const bool vis = (kv < Sk) && (!CAUSAL || kv <= qrow);
                        pv = vis ? pv : 0.f;
                    }
                    ds[r] = pv * (dp[r] - delta);
                }
                v8 dsb[2];
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    u32x4_t u;
#pragma unroll
                    for (int j = 0; j < 4; ++j) u[j] = T::pack2(ds[8 * kk + 2 * j], ds[8 * kk + 2 * j + 1]);
                    dsb[kk] = as_v8<T>(u);
                }
                // dQ^T += K^T . dS^T  (A = K^T by transpose read, B = dS in registers)
#pragma unroll
                for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                    for (int d = 0; d < DB; ++d) {
                        const int off = ((8 * sb + 4 * kk) * (D / 16) + 2 * d) * 128;
                        const s16x4_t a0 = lds_tr16(kst_img + off);
                        const s16x4_t a1 = lds_tr16(kst_img + off + 2 * (D / 16) * 128);
                        acc[d] = T::mfma(as_v8<T>(a0, a1), dsb[kk], acc[d]);
                    }
            }
        }
        if (t + 1 < nt) write_stage(cur ^ 1);
        __syncthreads();
    }

    if (qrow < Sq) {
        char* orow = reinterpret_cast<char*>(p.dq) + (qbase + qrow) * RB;
        const float sc = p.scale;
#pragma unroll
        for (int d = 0; d < DB; ++d)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4)
                store4<T>(orow, 32 * d + 8 * g4 + 4 * hi, acc[d][4 * g4] * sc, acc[d][4 * g4 + 1] * sc,
                          acc[d][4 * g4 + 2] * sc, acc[d][4 * g4 + 3] * sc);
    }
}

// ------------------------------------------------------------ dK/dV kernel ----
constexpr int kKvBlock = 128;  // 4 waves x 32 key rows
constexpr int kQT = 32;        // query rows per tile

template <int D>
struct DkvCfg {
    static constexpr int RB = D * 2, CPR = RB / 16, TILE = kQT * RB, NCHUNK = TILE / 16;
    static constexpr int CH = (NCHUNK + 255) / 256, KS = D / 16, DB = D / 32;
    // per stage: Q row-major, Q sub-tiled, dO row-major, dO sub-tiled, LSE*log2e[32], delta[32]
    static constexpr int STAGE = 4 * TILE + 256;
    static constexpr int LDS = 2 * STAGE;
};

template <class T, int D, bool CAUSAL>
__global__ void __launch_bounds__(256) fa_bwd_dkdv_kernel(const BwdParams p) {
    using Cfg = DkvCfg<D>;
    using v8 = typename T::v8;
    constexpr int RB = Cfg::RB, CPR = Cfg::CPR, TILE = Cfg::TILE, CH = Cfg::CH, KS = Cfg::KS, DB = Cfg::DB;
    constexpr int STAGE = Cfg::STAGE;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = p.Hq / p.Hkv;
    // one work item per (batch, kv head, kv block): reuse decode_work with Hq := Hkv
    const WorkItem w = decode_work(blockIdx.x, p.B, p.Hkv, p.Hkv, p.nblk, false);
    const int Sq = p.Sq, Sk = p.Sk;
    const int n0w = w.blk * kKvBlock + wave * 32;  // first key row of this wave
    const int kvrow = n0w + l31;
    const int kvr = kvrow < Sk ? kvrow : Sk - 1;
    const size_t kvbase = (size_t)(w.b * p.Hkv + w.hk) * Sk;

    v8 kf[KS], vf[KS];  // B operands: lane (kv, hi) holds d = 16ks + 8hi .. +7
    {
        const u32x4_t* kp = reinterpret_cast<const u32x4_t*>(p.k) + (kvbase + kvr) * CPR;
        const u32x4_t* vp = reinterpret_cast<const u32x4_t*>(p.v) + (kvbase + kvr) * CPR;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            kf[ks] = as_v8<T>(kp[2 * ks + hi]);
            vf[ks] = as_v8<T>(vp[2 * ks + hi]);
        }
    }
    const float c = p.c;

    int st_row[CH], st_cc[CH];
#pragma unroll
    for (int i = 0; i < CH; ++i) {
        const int cidx = tid + 256 * i;
        st_row[i] = cidx / CPR;
        st_cc[i] = cidx % CPR;
    }
    int a_off[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) a_off[ks] = rm_off<D>(l31, 2 * ks + hi);
    const int tr_off = hi * (D / 16) * 128 + ((lane >> 4) & 1) * 128 + (lane & 15) * 8;

    // query tiles that can see this KV block (top-left causal: q >= kv)
    const int ntq_all = (Sq + kQT - 1) / kQT;
    const int first_qt = CAUSAL ? (w.blk * kKvBlock) / kQT : 0;
    const int ntq = ntq_all > first_qt ? ntq_all - first_qt : 0;
    const int nit = ntq * g;  // flattened (group head, q tile) loop

    u32x4_t qst[CH], dst[CH];
    float sc_st = 0.f;  // staged LSE (threads 0..31) or delta (threads 32..63)
    auto issue_loads = [&](int it) {
        const int hh = it / ntq, qt = first_qt + it % ntq;
        const size_t qb = (size_t)(w.b * p.Hq + w.hk * g + hh) * Sq;
        const int q0 = qt * kQT;
#pragma unroll
        for (int i = 0; i < CH; ++i)
            if (Cfg::NCHUNK % 256 == 0 || tid + 256 * i < Cfg::NCHUNK) {
                int r = q0 + st_row[i];
                r = r < Sq ? r : Sq - 1;
                qst[i] = reinterpret_cast<const u32x4_t*>(p.q)[(qb + r) * CPR + st_cc[i]];
                dst[i] = reinterpret_cast<const u32x4_t*>(p.dout)[(qb + r) * CPR + st_cc[i]];
            }
        if (tid < 64) {
            int r = q0 + (tid & 31);
            r = r < Sq ? r : Sq - 1;
            sc_st = tid < 32 ? p.lse[qb + r] * kLog2e : p.delta[qb + r];
        }
    };
    auto write_stage = [&](int buf) {
        char* base = smem + buf * STAGE;
#pragma unroll
        for (int i = 0; i < CH; ++i)
            if (Cfg::NCHUNK % 256 == 0 || tid + 256 * i < Cfg::NCHUNK) {
                *reinterpret_cast<u32x4_t*>(base + rm_off<D>(st_row[i], st_cc[i])) = qst[i];
                *reinterpret_cast<u32x4_t*>(base + TILE + st_off<D>(st_row[i], st_cc[i])) = qst[i];
                *reinterpret_cast<u32x4_t*>(base + 2 * TILE + rm_off<D>(st_row[i], st_cc[i])) = dst[i];
                *reinterpret_cast<u32x4_t*>(base + 3 * TILE + st_off<D>(st_row[i], st_cc[i])) = dst[i];
            }
        if (tid < 64) reinterpret_cast<float*>(base + 4 * TILE)[tid] = sc_st;
    };

    f32x16_t dk[DB], dv[DB];
#pragma unroll
    for (int d = 0; d < DB; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) { dk[d][r] = 0.f; dv[d][r] = 0.f; }

    if (nit > 0) {
        issue_loads(0);
        write_stage(0);
    }
    __syncthreads();

    for (int it = 0; it < nit; ++it) {
        const int cur = it & 1;
        const int q0 = (first_qt + it % ntq) * kQT;
        if (it + 1 < nit) issue_loads(it + 1);
        // the tile contributes to this wave's keys iff some query row q >= key row exists
        if (!CAUSAL || q0 + kQT - 1 >= n0w) {
            const char* base = smem + cur * STAGE;
            const char* qrm = base;
            const char* qst_img = base + TILE + tr_off;
            const char* drm = base + 2 * TILE;
            const char* dst_img = base + 3 * TILE + tr_off;
            const float* scal = reinterpret_cast<const float*>(base + 4 * TILE);

            f32x16_t s, dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const u32x4_t qa = *reinterpret_cast<const u32x4_t*>(qrm + a_off[ks]);
                const u32x4_t da = *reinterpret_cast<const u32x4_t*>(drm + a_off[ks]);
                s = T::mfma(as_v8<T>(qa), kf[ks], s);    // S  = Q  . K^T   (rows q, cols kv)
                dp = T::mfma(as_v8<T>(da), vf[ks], dp);  // dP = dO . V^T
            }
            const bool need_mask = (CAUSAL && (q0 < n0w + 31)) || (q0 + kQT > Sq) || (n0w + 32 > Sk);
            float pr[16], ds[16];
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const f32x4_t l4 = *reinterpret_cast<const f32x4_t*>(scal + 8 * g4 + 4 * hi);
                const f32x4_t d4 = *reinterpret_cast<const f32x4_t*>(scal + 32 + 8 * g4 + 4 * hi);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int r = 4 * g4 + j;
                    float pv = fast_exp2(__builtin_fmaf(s[r], c, -l4[j]));
                    if (need_mask) {
                        const int q = q0 + crow(r, hi);
                        const bool vis = (q < Sq) && (kvrow < Sk) && (!CAUSAL || kvrow <= q);
                        pv = vis ? pv : 0.f;
                    }
                    pr[r] = pv;
                    ds[r] = pv * (dp[r] - d4[j]);
                }
            }
            v8 pb[2], dsb[2];
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                u32x4_t u, u2;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    u[j] = T::pack2(pr[8 * kk + 2 * j], pr[8 * kk + 2 * j + 1]);
                    u2[j] = T::pack2(ds[8 * kk + 2 * j], ds[8 * kk + 2 * j + 1]);
                }
                pb[kk] = as_v8<T>(u);
                dsb[kk] = as_v8<T>(u2);
            }
            // dV^T += dO^T . P ; dK^T += Q^T . dS   (A by transpose read, k-slot = query row)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int d = 0; d < DB; ++d) {
                    const int off = ((4 * kk) * (D / 16) + 2 * d) * 128;
                    const s16x4_t x0 = lds_tr16(dst_img + off);
                    const s16x4_t x1 = lds_tr16(dst_img + off + 2 * (D / 16) * 128);
                    dv[d] = T::mfma(as_v8<T>(x0, x1), pb[kk], dv[d]);
                    const s16x4_t y0 = lds_tr16(qst_img + off);
                    const s16x4_t y1 = lds_tr16(qst_img + off + 2 * (D / 16) * 128);
                    dk[d] = T::mfma(as_v8<T>(y0, y1), dsb[kk], dk[d]);
                }
        }
        if (it + 1 < nit) write_stage(cur ^ 1);
        __syncthreads();
    }

    if (kvrow < Sk) {
        char* krow = reinterpret_cast<char*>(p.dk) + (kvbase + kvrow) * RB;
        char* vrow = reinterpret_cast<char*>(p.dv) + (kvbase + kvrow) * RB;
        const float sc = p.scale;
#pragma unroll
        for (int d = 0; d < DB; ++d)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int de = 32 * d + 8 * g4 + 4 * hi;
                store4<T>(krow, de, dk[d][4 * g4] * sc, dk[d][4 * g4 + 1] * sc, dk[d][4 * g4 + 2] * sc,
                          dk[d][4 * g4 + 3] * sc);
                store4<T>(vrow, de, dv[d][4 * g4], dv[d][4 * g4 + 1], dv[d][4 * g4 + 2], dv[d][4 * g4 + 3]);
            }
    }
}

template <class T, int D>
int launch_bwd_16(const BwdArgs& a, hipStream_t stream) {
    {
        DeltaParams dp;
        dp.o = a.o; dp.dout = a.dout; dp.delta = a.delta;
        dp.rows = (long long)a.B * a.Hq * a.Sq;
        constexpr int RPB = 256 / (D * 2 / 16);
        const dim3 grid((unsigned)((dp.rows + RPB - 1) / RPB)), block(256);
        hipLaunchKernelGGL((fa_bwd_delta_kernel<T, D>), grid, block, 0, stream, dp);
        int rc = (int)hipGetLastError();
        if (rc) return rc;
    }
    BwdParams p;
    p.q = a.q; p.k = a.k; p.v = a.v; p.dout = a.dout; p.lse = a.lse; p.delta = a.delta;
    p.dq = a.dq; p.dk = a.dk; p.dv = a.dv;
    p.B = a.B; p.Hq = a.Hq; p.Hkv = a.Hkv; p.Sq = a.Sq; p.Sk = a.Sk;
    p.c = a.scale * kLog2e;
    p.scale = a.scale;
    {
        p.nblk = (a.Sq + kDqQBlock - 1) / kDqQBlock;
        const dim3 grid((unsigned)(p.nblk * a.B * a.Hq)), block(512);
        if (a.causal)
            hipLaunchKernelGGL((fa_bwd_dq_kernel<T, D, true>), grid, block, DqCfg<D>::LDS, stream, p);
        else
            hipLaunchKernelGGL((fa_bwd_dq_kernel<T, D, false>), grid, block, DqCfg<D>::LDS, stream, p);
        int rc = (int)hipGetLastError();
        if (rc) return rc;
    }
    {
        p.nblk = (a.Sk + kKvBlock - 1) / kKvBlock;
        const dim3 grid((unsigned)(p.nblk * a.B * a.Hkv)), block(256);
        if (a.causal)
            hipLaunchKernelGGL((fa_bwd_dkdv_kernel<T, D, true>), grid, block, DkvCfg<D>::LDS, stream, p);
        else
            hipLaunchKernelGGL((fa_bwd_dkdv_kernel<T, D, false>), grid, block, DkvCfg<D>::LDS, stream, p);
        return (int)hipGetLastError();
    }
}

template <class T, int D>
int set_attr_bwd() {
    int rc = 0;
    rc |= (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&fa_bwd_dq_kernel<T, D, true>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, DqCfg<D>::LDS);
    rc |= (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&fa_bwd_dq_kernel<T, D, false>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, DqCfg<D>::LDS);
    rc |= (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&fa_bwd_dkdv_kernel<T, D, true>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, DkvCfg<D>::LDS);
    rc |= (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&fa_bwd_dkdv_kernel<T, D, false>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, DkvCfg<D>::LDS);
    return rc;
}

}  // namespace

int launch_bwd_f32(const BwdArgs& a, hipStream_t stream);  // fa_bwd_f32.hip
int configure_bwd_f32();

// Shared with fa_bwd_f32.hip
int launch_delta_f32(const BwdArgs& a, hipStream_t stream) {
    DeltaParams dp;
    dp.o = a.o; dp.dout = a.dout; dp.delta = a.delta;
    dp.rows = (long long)a.B * a.Hq * a.Sq;
    const dim3 block(256);
    if (a.D == 128) {
        const dim3 grid((unsigned)((dp.rows + 7) / 8));
        hipLaunchKernelGGL((fa_bwd_delta_f32_kernel<128>), grid, block, 0, stream, dp);
    } else if (a.D == 64) {
        const dim3 grid((unsigned)((dp.rows + 15) / 16));
        hipLaunchKernelGGL((fa_bwd_delta_f32_kernel<64>), grid, block, 0, stream, dp);
    } else if (a.D == 32) {
        const dim3 grid((unsigned)((dp.rows + 31) / 32));
        hipLaunchKernelGGL((fa_bwd_delta_f32_kernel<32>), grid, block, 0, stream, dp);
    } else {
        return -1;
    }
    return (int)hipGetLastError();
}

uint64_t bwd_workspace_bytes(int B, int Hq, int Sq) { return (uint64_t)B * Hq * Sq * sizeof(float); }

int launch_bwd(const BwdArgs& a, hipStream_t stream) {
    if (a.dtype == kF32) return launch_bwd_f32(a, stream);
    if (a.dtype == kBF16) {
        if (a.D == 128) return launch_bwd_16<Bf16Traits, 128>(a, stream);
        if (a.D == 64) return launch_bwd_16<Bf16Traits, 64>(a, stream);
        if (a.D == 32) return launch_bwd_16<Bf16Traits, 32>(a, stream);
    } else if (a.dtype == kF16) {
        if (a.D == 128) return launch_bwd_16<F16Traits, 128>(a, stream);
        if (a.D == 64) return launch_bwd_16<F16Traits, 64>(a, stream);
        if (a.D == 32) return launch_bwd_16<F16Traits, 32>(a, stream);
    }
    return -1;
}

int configure_bwd() {
    int rc = 0;
    rc |= set_attr_bwd<Bf16Traits, 128>();
    rc |= set_attr_bwd<Bf16Traits, 64>();
    rc |= set_attr_bwd<Bf16Traits, 32>();
    rc |= set_attr_bwd<F16Traits, 128>();
    rc |= set_attr_bwd<F16Traits, 64>();
    rc |= set_attr_bwd<F16Traits, 32>();
    rc |= configure_bwd_f32();
    return rc;
}

}  // namespace aule_hip
