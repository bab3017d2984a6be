// fa_bwd_f32.hip -- fp32 FlashAttention backward for gfx950 (exact-f32 MFMA).
//
// Arithmetic behind the legacy aule_attention_backward (src/lib.zig:639-762 ->
// src/attention_backward_pipeline.zig:472-537 -> shaders/attention_backward_f32.comp)
// and behind autograd for fp32 tensors.  Same three-launch structure as the 16-bit
// backward (fa_bwd_gfx950.hip): delta, dQ (lane owns a query row), dK/dV (lane owns a
// key row, GQA group reduced by looping the group's query heads).  All tile products use
// v_mfma_f32_32x32x2_f32 (f32 in, f32 accumulate).
//
// MFMA 32x32x2 operand map: A[i = lane&31][k = lane>>5], B[k = lane>>5][n = lane&31];
// step r of a P/dS product contracts the row pair {crow(r,0), crow(r,1)} = {x, x+4},
// which is exactly what accumulator register r of the two lane halves holds.
#include <cstdlib>

#include "fa_device.h"
#include "fa_kernels.h"

namespace aule_hip {

int launch_delta_f32(const BwdArgs& a, hipStream_t stream);  // fa_bwd_gfx950.hip

namespace {

struct BwdF32Params {
    const float* q;
    const float* k;
    const float* v;
    const float* dout;
    const float* lse;
    const float* delta;
    float* dq;
    float* dk;
    float* dv;
    int B, Hq, Hkv, Sq, Sk;
    float c;      // scale * log2(e)
    float scale;
    int nblk;
    int window;   // sliding window: key j visible to query i only if i - j < window (0: off)
    int coff;     // causal position offset (query i sits at position i + coff)
    // small grids (round 5): npiece work items per block, each with 1 / npiece of the block's tiles (dQ: key tiles; dK/dV: the query tiles of
    // every head of the group); a piece writes its (scaled) sums to part[piece][...] and fa_bwd_f32_sum adds the pieces in a fixed order
    int npiece;
    float* part;      // dQ kernel: [npiece][B Hq Sq][D]; dK/dV kernel: [npiece][2][B Hkv Sk][D] (dK, dV)
    long long prows;  // rows per piece plane
};

constexpr int kRows = 128;  // rows per workgroup (4 waves x 32)
constexpr int kTile = 32;

// Staging of a [32 x D] fp32 tile whose rows start at `row0` (clamped to nrows-1), round 4.
//
// Two LDS images, both written from ONE set of registers (thread -> row cidx & 31, four columns 4 (cidx >> 5) .. + 3):
//   tq : the A operands of the products that contract over D (S = X Y^T): lane (row, hi) needs X[row][2 st + hi] for st = 0 .. D/2 - 1.
//        Element (row, d) sits at tq[(((d >> 3) * 2 + (d & 1)) * 32 + row) * 4 + ((d >> 1) & 3)]: ONE ds_read_b128 hands a lane its
//        operands of four consecutive MFMAs, 16 lanes of a pass read 256 contiguous bytes (no bank conflict).
//   cm : the A operands of the products that contract over the tile's ROWS (dQ += dS K, dV += P^T dO, dK += dS^T Q): lane (col, hi)
//        needs X[crow(r, hi)][col], r = 0 .. 15, i.e. four runs of four consecutive rows.  Column-major with a pitch of 36 floats
//        (cm[col * 36 + row]): one ds_read_b128 per run, and the 16 lanes of a pass (pitch 36 = 9 x 16 bytes) cover all 64 banks once.
// (Rounds 1-3: a [D][32] image and a row-major one, filled by two separate global loads per element and read with one ds_read_b32
//  per MFMA -- 192-256 LDS instructions and as many fine-grained waits per tile.)
// tile_load issues the global loads of a tile into registers, tile_store writes them to LDS: the kernels load tile t + 1 BEFORE
// they compute tile t and store it behind the tile's closing barrier, so the global-memory latency runs under a tile's MFMAs
// (the staging used to be load -> store -> barrier at the top of every tile, with the matrix pipes idle).
constexpr int kCmPitch = 36;
template <int D>
struct TileRegs {
    static constexpr int NCH = kTile * (D / 4) / 256;
    f32x4_t x[NCH];
};
template <int D>
__device__ __forceinline__ void tile_load(TileRegs<D>& t, const float* __restrict__ g, int row0, int nrows, int tid) {
#pragma unroll
    for (int i = 0; i < TileRegs<D>::NCH; ++i) {
        const int cidx = tid + 256 * i, rr = cidx & 31, dc = cidx >> 5;
        int r = row0 + rr;
        r = r < nrows ? r : nrows - 1;
        t.x[i] = *reinterpret_cast<const f32x4_t*>(g + (size_t)r * D + 4 * dc);
    }
}
template <int D>
__device__ __forceinline__ void tile_store(const TileRegs<D>& t, float* tq, float* cm, int tid) {
#pragma unroll
    for (int i = 0; i < TileRegs<D>::NCH; ++i) {
        const int cidx = tid + 256 * i, rr = cidx & 31, dc = cidx >> 5;
        const f32x4_t x = t.x[i];
        if (tq != nullptr) {   // d = 4 dc + e: group dc >> 1, parity e & 1, slot 2 (dc & 1) + (e >> 1)
            float* const q0 = tq + (((dc >> 1) * 2 + 0) * 32 + rr) * 4 + 2 * (dc & 1);
            float* const q1 = tq + (((dc >> 1) * 2 + 1) * 32 + rr) * 4 + 2 * (dc & 1);
            *reinterpret_cast<f32x2_t*>(q0) = f32x2_t{x[0], x[2]};
            *reinterpret_cast<f32x2_t*>(q1) = f32x2_t{x[1], x[3]};
        }
        if (cm != nullptr) {
            cm[(4 * dc + 0) * kCmPitch + rr] = x[0];
            cm[(4 * dc + 1) * kCmPitch + rr] = x[1];
            cm[(4 * dc + 2) * kCmPitch + rr] = x[2];
            cm[(4 * dc + 3) * kCmPitch + rr] = x[3];
        }
    }
}
// the four A operands of MFMAs 4 st4 .. 4 st4 + 3 of a product over D
__device__ __forceinline__ f32x4_t tq_read(const float* tq, int st4, int hi, int l31) {
    return *reinterpret_cast<const f32x4_t*>(tq + ((st4 * 2 + hi) * 32 + l31) * 4);
}
// the four A operands of steps r = 4 q .. 4 q + 3 of a product over the rows, column `col`
__device__ __forceinline__ f32x4_t cm_read(const float* cm, int col, int q, int hi) {
    return *reinterpret_cast<const f32x4_t*>(cm + col * kCmPitch + 8 * q + 4 * hi);
}

// lane (row, hi) loads row[2s + hi] for s = 0..D/2-1 (B operand of the 32x32x2 MFMA)
template <int D>
__device__ __forceinline__ void load_b_operand(const float* rowp, int hi, float (&f)[D / 2]) {
#pragma unroll
    for (int s4 = 0; s4 < D / 4; ++s4) {
        const f32x4_t x = *reinterpret_cast<const f32x4_t*>(rowp + 4 * s4);
        f[2 * s4] = hi ? x[1] : x[0];
        f[2 * s4 + 1] = hi ? x[3] : x[2];
    }
}

template <int D, bool CAUSAL>
__global__ void __launch_bounds__(256, D <= 64 ? 2 : 1) fa_bwd_dq_f32_kernel(const BwdF32Params p) {
    constexpr int DB = D / 32;
    __shared__ __attribute__((aligned(16))) float Kt[D * 32];          // tq image of the K tile
    __shared__ __attribute__((aligned(16))) float Kcm[D * kCmPitch];   // cm image of the K tile
    __shared__ __attribute__((aligned(16))) float Vt[D * 32];          // tq image of the V tile

    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int piece = p.npiece > 1 ? (int)(blockIdx.x % (unsigned)p.npiece) : 0;
    const int item = p.npiece > 1 ? (int)(blockIdx.x / (unsigned)p.npiece) : (int)blockIdx.x;
    const WorkItem w = decode_work_ranked(item, p.B, p.Hq, p.Hkv, p.nblk, CAUSAL);   // (causal: every unit's last block first)
    const int Sq = p.Sq, Sk = p.Sk;
    const int q0w = w.blk * kRows + wave * 32;
    const int qrow = q0w + l31;
    const int qr = qrow < Sq ? qrow : Sq - 1;
    const size_t qbase = (size_t)(w.b * p.Hq + w.h) * Sq;
    const float* __restrict__ kg = p.k + (size_t)(w.b * p.Hkv + w.hk) * Sk * D;
    const float* __restrict__ vg = p.v + (size_t)(w.b * p.Hkv + w.hk) * Sk * D;

    float qf[D / 2], dof[D / 2];
    load_b_operand<D>(p.q + (qbase + qr) * D, hi, qf);
    load_b_operand<D>(p.dout + (qbase + qr) * D, hi, dof);
    const float lse2 = p.lse[qbase + qr] * kLog2e;
    const float delta = p.delta[qbase + qr];
    const float c = p.c;

    f32x16_t acc[DB];
#pragma unroll
    for (int d = 0; d < DB; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[d][r] = 0.f;

    const int coff = p.coff;
    const int kv_hi = CAUSAL ? min(Sk, w.blk * kRows + kRows + coff) : Sk;
    const int nt = (kv_hi + kTile - 1) / kTile;
    const int wave_kv_hi = CAUSAL ? min(Sk, q0w + 32 + coff) : Sk;
    const int W = p.window;
    int t_lo = W > 0 ? max(0, w.blk * kRows + coff - W + 1) / kTile : 0;  // tiles before the block's window: skipped
    const int wave_kv_lo = W > 0 ? q0w + coff - W + 1 : 0;
    int nt_end = nt;
    if (p.npiece > 1) {   // this piece's share of the block's key tiles
        const int chunk = (max(0, nt - t_lo) + p.npiece - 1) / p.npiece;
        t_lo = t_lo + piece * chunk;
        nt_end = min(nt, t_lo + chunk);
    }

    TileRegs<D> kreg, vreg;
    if (t_lo < nt_end) {
        tile_load(kreg, kg, t_lo * kTile, Sk, tid);
        tile_load(vreg, vg, t_lo * kTile, Sk, tid);
    }
    for (int t = t_lo; t < nt_end; ++t) {
        const int kv0 = t * kTile;
        tile_store(kreg, Kt, Kcm, tid);
        tile_store(vreg, Vt, static_cast<float*>(nullptr), tid);
        __syncthreads();
        if (t + 1 < nt_end) {   // the next tile's rows travel while this one is computed
            tile_load(kreg, kg, kv0 + kTile, Sk, tid);
            tile_load(vreg, vg, kv0 + kTile, Sk, tid);
        }
        if (kv0 < wave_kv_hi && kv0 + kTile > wave_kv_lo) {
            f32x16_t s, dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
            for (int st4 = 0; st4 < D / 8; ++st4) {
                const f32x4_t ka = tq_read(Kt, st4, hi, l31), va = tq_read(Vt, st4, hi, l31);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    s = __builtin_amdgcn_mfma_f32_32x32x2f32(ka[j], qf[4 * st4 + j], s, 0, 0, 0);
                    dp = __builtin_amdgcn_mfma_f32_32x32x2f32(va[j], dof[4 * st4 + j], dp, 0, 0, 0);
                }
            }
            const bool need_mask = (CAUSAL && (kv0 + kTile - 1 > q0w + coff)) || (kv0 + kTile > Sk) || (W > 0 && q0w + coff + 31 - kv0 >= W);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float pv = fast_exp2(__builtin_fmaf(s[r], c, -lse2));
                if (need_mask) {
                    const int kv = kv0 + crow(r, hi);
                    const bool vis = (kv < Sk) && (!CAUSAL || kv <= qrow + coff) && (W <= 0 || qrow + coff - kv < W);
                    pv = vis ? pv : 0.f;
                }
                s[r] = pv * (dp[r] - delta);  // dS^T (scale applied in the epilogue)
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x4_t ka[DB];
#pragma unroll
                for (int d = 0; d < DB; ++d) ka[d] = cm_read(Kcm, 32 * d + l31, q, hi);
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int d = 0; d < DB; ++d)
                        acc[d] = __builtin_amdgcn_mfma_f32_32x32x2f32(ka[d][j], s[4 * q + j], acc[d], 0, 0, 0);
            }
        }
        __syncthreads();
    }

    if (qrow < Sq) {
        float* orow = (p.npiece > 1 ? p.part + (size_t)piece * p.prows * D : p.dq) + (qbase + qrow) * D;
        const float sc = p.scale;
#pragma unroll
        for (int d = 0; d < DB; ++d)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                f32x4_t x = {acc[d][4 * g4] * sc, acc[d][4 * g4 + 1] * sc, acc[d][4 * g4 + 2] * sc,
                             acc[d][4 * g4 + 3] * sc};
                *reinterpret_cast<f32x4_t*>(orow + 32 * d + 8 * g4 + 4 * hi) = x;
            }
    }
}

template <int D>
struct DkvF32Cfg {
    static constexpr int IMG = kTile * D;              // floats per tq image
    static constexpr int CMG = kCmPitch * D;           // floats per cm image
    static constexpr int LDS = (2 * IMG + 2 * CMG + 64) * 4;     // Q tq, Q cm, dO tq, dO cm, scal[64]
};

template <int D, bool CAUSAL>
__global__ void __launch_bounds__(256, D <= 64 ? 2 : 1) fa_bwd_dkdv_f32_kernel(const BwdF32Params p) {
    constexpr int DB = D / 32, IMG = DkvF32Cfg<D>::IMG, CMG = DkvF32Cfg<D>::CMG;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* const Qt = reinterpret_cast<float*>(smem_raw);
    float* const Qcm = Qt + IMG;
    float* const Gt = Qcm + CMG;
    float* const Gcm = Gt + IMG;
    float* const scal = Gcm + CMG;

    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = p.Hq / p.Hkv;
    const int piece = p.npiece > 1 ? (int)(blockIdx.x % (unsigned)p.npiece) : 0;
    const int item = p.npiece > 1 ? (int)(blockIdx.x / (unsigned)p.npiece) : (int)blockIdx.x;
    const WorkItem w = decode_work_ranked(item, p.B, p.Hkv, p.Hkv, p.nblk, false);   // (key block 0 sees the most queries: first)
    const int Sq = p.Sq, Sk = p.Sk;
    const int n0w = w.blk * kRows + wave * 32;
    const int kvrow = n0w + l31;
    const int kvr = kvrow < Sk ? kvrow : Sk - 1;
    const size_t kvbase = (size_t)(w.b * p.Hkv + w.hk) * Sk;

    float kf[D / 2], vf[D / 2];
    load_b_operand<D>(p.k + (kvbase + kvr) * D, hi, kf);
    load_b_operand<D>(p.v + (kvbase + kvr) * D, hi, vf);
    const float c = p.c;

    f32x16_t dk[DB], dv[DB];
#pragma unroll
    for (int d = 0; d < DB; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) { dk[d][r] = 0.f; dv[d][r] = 0.f; }

    const int W = p.window;
    const int coff = p.coff;   // query q sits at position q + coff
    int ntq_all = (Sq + kTile - 1) / kTile;
    if (W > 0) ntq_all = min(ntq_all, max(0, w.blk * kRows + kRows - 1 + W - coff + kTile - 1) / kTile);  // q + coff - kv < W
    int first_qt = CAUSAL ? max(0, w.blk * kRows - coff) / kTile : 0;
    if (p.npiece > 1) {   // this piece's share of the query tiles (of every head of the group)
        const int chunk = (max(0, ntq_all - first_qt) + p.npiece - 1) / p.npiece;
        first_qt = first_qt + piece * chunk;
        ntq_all = min(ntq_all, first_qt + chunk);
    }

    for (int hh = 0; hh < g; ++hh) {
        const size_t qb = (size_t)(w.b * p.Hq + w.hk * g + hh) * Sq;
        TileRegs<D> qreg, greg;
        float sreg = 0.f;
        auto load_block = [&](int q0) __attribute__((always_inline)) {
            tile_load(qreg, p.q + qb * D, q0, Sq, tid);
            tile_load(greg, p.dout + qb * D, q0, Sq, tid);
            if (tid < 64) {
                int r = q0 + (tid & 31);
                r = r < Sq ? r : Sq - 1;
                sreg = tid < 32 ? p.lse[qb + r] * kLog2e : p.delta[qb + r];
            }
        };
        if (first_qt < ntq_all) load_block(first_qt * kTile);
        for (int qt = first_qt; qt < ntq_all; ++qt) {
            const int q0 = qt * kTile;
            tile_store(qreg, Qt, Qcm, tid);
            tile_store(greg, Gt, Gcm, tid);
            if (tid < 64) scal[tid] = sreg;
            __syncthreads();
            if (qt + 1 < ntq_all) load_block(q0 + kTile);   // the next query block travels while this one is computed
            if ((!CAUSAL || q0 + coff + kTile - 1 >= n0w) && (W <= 0 || q0 + coff < n0w + 31 + W)) {
                f32x16_t s, dp;
#pragma unroll
                for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
                for (int st4 = 0; st4 < D / 8; ++st4) {
                    const f32x4_t qa = tq_read(Qt, st4, hi, l31), ga = tq_read(Gt, st4, hi, l31);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        s = __builtin_amdgcn_mfma_f32_32x32x2f32(qa[j], kf[4 * st4 + j], s, 0, 0, 0);
                        dp = __builtin_amdgcn_mfma_f32_32x32x2f32(ga[j], vf[4 * st4 + j], dp, 0, 0, 0);
                    }
                }
                const bool need_mask = (CAUSAL && (q0 + coff < n0w + 31)) || (q0 + kTile > Sq) || (n0w + 32 > Sk) || (W > 0 && q0 + coff + kTile - 1 - n0w >= W);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ql = crow(r, hi);
                    float pv = fast_exp2(__builtin_fmaf(s[r], c, -scal[ql]));
                    if (need_mask) {
                        const int q = q0 + ql;
                        const bool vis = (q < Sq) && (kvrow < Sk) && (!CAUSAL || kvrow <= q + coff) && (W <= 0 || q + coff - kvrow < W);
                        pv = vis ? pv : 0.f;
                    }
                    s[r] = pv;                              // P
                    dp[r] = pv * (dp[r] - scal[32 + ql]);   // dS (unscaled)
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    f32x4_t ga[DB], qa[DB];
#pragma unroll
                    for (int d = 0; d < DB; ++d) {
                        ga[d] = cm_read(Gcm, 32 * d + l31, q, hi);
                        qa[d] = cm_read(Qcm, 32 * d + l31, q, hi);
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int d = 0; d < DB; ++d) {
                            dv[d] = __builtin_amdgcn_mfma_f32_32x32x2f32(ga[d][j], s[4 * q + j], dv[d], 0, 0, 0);
                            dk[d] = __builtin_amdgcn_mfma_f32_32x32x2f32(qa[d][j], dp[4 * q + j], dk[d], 0, 0, 0);
                        }
                }
            }
            __syncthreads();
        }
    }

    if (kvrow < Sk) {
        float* krow = (p.npiece > 1 ? p.part + (size_t)(2 * piece) * p.prows * D : p.dk) + (kvbase + kvrow) * D;
        float* vrow = (p.npiece > 1 ? p.part + (size_t)(2 * piece + 1) * p.prows * D : p.dv) + (kvbase + kvrow) * D;
        const float sc = p.scale;
#pragma unroll
        for (int d = 0; d < DB; ++d)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int de = 32 * d + 8 * g4 + 4 * hi;
                f32x4_t x = {dk[d][4 * g4] * sc, dk[d][4 * g4 + 1] * sc, dk[d][4 * g4 + 2] * sc,
                             dk[d][4 * g4 + 3] * sc};
                f32x4_t y = {dv[d][4 * g4], dv[d][4 * g4 + 1], dv[d][4 * g4 + 2], dv[d][4 * g4 + 3]};
                *reinterpret_cast<f32x4_t*>(krow + de) = x;
                *reinterpret_cast<f32x4_t*>(vrow + de) = y;
            }
    }
}

// out[i] = sum over the pieces, in piece order (deterministic): n4 float4 per plane, `nout` planes per piece (dQ: 1; dK, dV: 2)
struct SumF32Params {
    const float* part;
    float* out0;
    float* out1;
    long long n4;
    int npiece, nout;
};
__global__ void __launch_bounds__(256) fa_bwd_f32_sum(const SumF32Params p) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= p.nout * p.n4) return;
    const int which = (int)(i / p.n4);
    const long long e = i % p.n4;
    const f32x4_t* src = reinterpret_cast<const f32x4_t*>(p.part) + (long long)which * p.n4 + e;
    f32x4_t acc = src[0];
    for (int j = 1; j < p.npiece; ++j) {
        const f32x4_t x = src[(long long)j * p.nout * p.n4];
        acc[0] += x[0]; acc[1] += x[1]; acc[2] += x[2]; acc[3] += x[3];
    }
    reinterpret_cast<f32x4_t*>(which ? p.out1 : p.out0)[e] = acc;
}

// Pieces per block for grids that leave most of the chip idle (the reference's Zig benchmark shape, tests/benchmark_attention.zig:18-21:
// B4 H8 S512 D64 = 128 workgroups for 512 slots): as many as fill the slots, at least two tiles each, at most 8.  AULE_HIP_F32_SPLIT=0: off.
inline int f32_bwd_pieces(long long items, int tiles, int D, int device) {
    static const int on = [] {
        const char* e = std::getenv("AULE_HIP_F32_SPLIT");
        return (e != nullptr && e[0] == '0') ? 0 : 1;
    }();
    if (!on) return 1;
    const long long slots = (D <= 64 ? 2LL : 1LL) * device_cu_count(device);
    if (items <= 0 || items * 2 > slots) return 1;
    long long n = slots / items;
    if (n > tiles / 2) n = tiles / 2;
    if (n > 8) n = 8;
    return n < 2 ? 1 : (int)n;
}
inline void f32_bwd_plan(int B, int Hq, int Hkv, int Sq, int Sk, int D, int causal, int coff, int device, int& nq, int& nk) {
    const int nblq = (Sq + kRows - 1) / kRows, nblk = (Sk + kRows - 1) / kRows;
    const int kv_hi = causal ? (Sk < Sq + coff ? Sk : Sq + coff) : Sk;
    nq = f32_bwd_pieces((long long)nblq * B * Hq, (kv_hi + kTile - 1) / kTile, D, device);
    nk = f32_bwd_pieces((long long)nblk * B * Hkv, (Sq + kTile - 1) / kTile, D, device);
}
inline uint64_t f32_delta_bytes(int B, int Hq, int Sq) { return (((uint64_t)B * Hq * Sq * sizeof(float)) + 255) / 256 * 256; }

template <int D>
int launch_bwd_f32_d(const BwdArgs& a, hipStream_t stream) {
    int rc = launch_delta_f32(a, stream);
    if (rc) return rc;
    BwdF32Params p;
    p.q = (const float*)a.q; p.k = (const float*)a.k; p.v = (const float*)a.v;
    p.dout = (const float*)a.dout; p.lse = a.lse; p.delta = a.delta;
    p.dq = (float*)a.dq; p.dk = (float*)a.dk; p.dv = (float*)a.dv;
    p.B = a.B; p.Hq = a.Hq; p.Hkv = a.Hkv; p.Sq = a.Sq; p.Sk = a.Sk;
    p.c = a.scale * kLog2e;
    p.scale = a.scale;
    p.window = a.window > 0 ? a.window : 0;
    p.coff = a.causal ? a.coff : 0;
    int nq = 1, nk = 1;
    // (a window changes the tile ranges per block: the plan keeps to the window-less count, pieces beyond a block's tiles are empty)
    f32_bwd_plan(a.B, a.Hq, a.Hkv, a.Sq, a.Sk, D, a.causal, p.coff, a.device, nq, nk);
    if (a.ws_bytes != 0) {
        // (the plan must never outgrow the bytes the caller was told to bring -- the size query and this launch can see
        // different CU counts if the caller's current device differs from the descriptor's and an entry point forgot to say so)
        const uint64_t have = a.ws_bytes > f32_delta_bytes(a.B, a.Hq, a.Sq) ? a.ws_bytes - f32_delta_bytes(a.B, a.Hq, a.Sq) : 0;
        const uint64_t per_q = (uint64_t)a.B * a.Hq * a.Sq * D * 4, per_k = 2ull * a.B * a.Hkv * a.Sk * D * 4;
        while (nq > 1 && (uint64_t)nq * per_q > have) --nq;
        while (nk > 1 && (uint64_t)nk * per_k > have) --nk;
    }
    float* const parts = reinterpret_cast<float*>(reinterpret_cast<char*>(a.delta) + f32_delta_bytes(a.B, a.Hq, a.Sq));   // (bwd_f32_partial_bytes() behind delta)
    const dim3 block(256);
    {
        p.nblk = (a.Sq + kRows - 1) / kRows;
        p.npiece = nq; p.part = parts; p.prows = (long long)a.B * a.Hq * a.Sq;
        const dim3 grid((unsigned)(p.nblk * a.B * a.Hq * nq));
        if (a.causal)
            hipLaunchKernelGGL((fa_bwd_dq_f32_kernel<D, true>), grid, block, 0, stream, p);
        else
            hipLaunchKernelGGL((fa_bwd_dq_f32_kernel<D, false>), grid, block, 0, stream, p);
        rc = (int)hipGetLastError();
        if (rc) return rc;
        if (nq > 1) {
            SumF32Params sp;
            sp.part = parts; sp.out0 = p.dq; sp.out1 = nullptr; sp.n4 = p.prows * D / 4; sp.npiece = nq; sp.nout = 1;
            hipLaunchKernelGGL(fa_bwd_f32_sum, dim3((unsigned)((sp.n4 + 255) / 256)), dim3(256), 0, stream, sp);
            rc = (int)hipGetLastError();
            if (rc) return rc;
        }
    }
    {
        p.nblk = (a.Sk + kRows - 1) / kRows;
        p.npiece = nk; p.part = parts; p.prows = (long long)a.B * a.Hkv * a.Sk;   // (the dQ pieces were summed: stream order)
        const dim3 grid((unsigned)(p.nblk * a.B * a.Hkv * nk));
        const size_t lds = DkvF32Cfg<D>::LDS;
        if (a.causal)
            hipLaunchKernelGGL((fa_bwd_dkdv_f32_kernel<D, true>), grid, block, lds, stream, p);
        else
            hipLaunchKernelGGL((fa_bwd_dkdv_f32_kernel<D, false>), grid, block, lds, stream, p);
        rc = (int)hipGetLastError();
        if (rc || nk == 1) return rc;
        SumF32Params sp;
        sp.part = parts; sp.out0 = p.dk; sp.out1 = p.dv; sp.n4 = p.prows * D / 4; sp.npiece = nk; sp.nout = 2;
        hipLaunchKernelGGL(fa_bwd_f32_sum, dim3((unsigned)((2 * sp.n4 + 255) / 256)), dim3(256), 0, stream, sp);
        return (int)hipGetLastError();
    }
}

template <int D>
int set_attr_f32() {
    int rc = (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&fa_bwd_dkdv_f32_kernel<D, true>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, DkvF32Cfg<D>::LDS);
    rc |= (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&fa_bwd_dkdv_f32_kernel<D, false>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, DkvF32Cfg<D>::LDS);
    return rc;
}

}  // namespace

// bytes of fp32 partial planes the small-grid split needs behind delta (0: no split for these sizes)
uint64_t bwd_f32_partial_bytes(int B, int Hq, int Hkv, int Sq, int Sk, int D, int causal, int device) {
    int nq = 1, nk = 1;
    f32_bwd_plan(B, Hq, Hkv, Sq, Sk, D, causal, 0, device, nq, nk);
    // (coff only shrinks the causal tile count the plan looks at: coff = Sk - Sq >= 0 gives at least as many tiles; size for both)
    int nq2 = 1, nk2 = 1;
    f32_bwd_plan(B, Hq, Hkv, Sq, Sk, D, causal, Sk > Sq ? Sk - Sq : 0, device, nq2, nk2);
    const uint64_t nqm = nq > nq2 ? nq : nq2, nkm = nk > nk2 ? nk : nk2;
    const uint64_t a = nqm > 1 ? nqm * (uint64_t)B * Hq * Sq * D * 4 : 0, b = nkm > 1 ? 2 * nkm * (uint64_t)B * Hkv * Sk * D * 4 : 0;
    return a > b ? a : b;
}

int launch_bwd_f32(const BwdArgs& a, hipStream_t stream) {
    if (a.D == 128) return launch_bwd_f32_d<128>(a, stream);
    if (a.D == 64) return launch_bwd_f32_d<64>(a, stream);
    if (a.D == 32) return launch_bwd_f32_d<32>(a, stream);
    return -1;
}

int configure_bwd_f32() { return set_attr_f32<128>() | set_attr_f32<64>() | set_attr_f32<32>(); }

}  // namespace aule_hip
