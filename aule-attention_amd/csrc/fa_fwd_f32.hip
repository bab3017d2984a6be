// fa_fwd_f32.hip -- fp32 FlashAttention forward for gfx950 (exact-f32 MFMA).
//
// This is the arithmetic behind the LEGACY fp32 C-ABI (aule_attention_forward,
// aule_attention_forward_gpu, aule_attention_forward_with_lse: src/lib.zig:312,
// :496, :765) and behind aule.flash_attention for fp32 torch / NumPy inputs; it
// takes the slot of shaders/attention_f32.comp / attention_forward_f32.comp and of
// src/backends/attention_hip.cpp.  Parity bar is 1e-5 (BASELINE.json), so both
// matmuls use v_mfma_f32_32x32x2_f32 (f32 in, f32 accumulate: bitwise an fmaf
// chain) rather than a 16-bit MFMA.
//
// Same lane-local structure as the 16-bit kernel (fa_fwd_gfx950.hip):
//   S^T[kv][q] = K.Q^T (A = K from LDS, B = Q in registers), lane owns q = lane&31;
//   O^T[d][q] = V^T.P^T with P's accumulator layout reused as the k-slot order.
// workgroup = 4 waves x 32 query rows, KV tile = 32 rows, single-buffered LDS with the next tile prefetched into
// registers, at most 256 VGPRs so that two workgroups share a CU:
//   Kt[d][32]  (transposed so that the A-operand ds_read_b32 is conflict-free)
//   V [32][D]  row-major.
#include <cstdlib>

#include "fa_device.h"
#include "fa_kernels.h"

namespace aule_hip {
namespace {

struct FwdF32Params {
    const float* q;
    const float* k;
    const float* v;
    float* o;
    float* lse;
    int B, Hq, Hkv, Sq, Sk;
    float c;   // scale * log2(e)  (sign kept: the max is taken on c*s)
    int nqb;
    int window;  // sliding window: key j visible to query i only if i - j < window (0: off)
    int coff;    // causal position offset (query i sits at position i + coff)
    // small grids (round 5): every Q block as npiece work items of 1 / npiece of its key tiles; a piece stores its un-normalised O row,
    // its running maximum (log2 units) and its row sum to part[piece][row][D + 4] and fa_fwd_f32_combine merges them
    int npiece;
    float* part;
    long long rows;   // B * Hq * Sq
};

constexpr int kQB = 128;  // 4 waves x 32 rows
constexpr int kKV = 32;

template <int D, bool CAUSAL>
__global__ void __launch_bounds__(256, 2) fa_fwd_f32_kernel(const FwdF32Params p) {   // (<= 256 registers: two workgroups per CU hide each other's staging latency; D = 128 took 276 and ran alone)
    constexpr int DB = D / 32;
    constexpr int C4 = D / 4;               // 16-byte chunks per row
    constexpr int NCH = kKV * C4 / 256;     // chunks per thread per tile (D=128: 4, 64: 2, 32: 1)
    __shared__ __attribute__((aligned(16))) float Kt[D * 32];
    __shared__ __attribute__((aligned(16))) float Vs[kKV * D];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;

    const int piece = p.npiece > 1 ? (int)(blockIdx.x % (unsigned)p.npiece) : 0;
    const int item = p.npiece > 1 ? (int)(blockIdx.x / (unsigned)p.npiece) : (int)blockIdx.x;
    const WorkItem w = decode_work_ranked(item, p.B, p.Hq, p.Hkv, p.nqb, CAUSAL);   // (causal: every unit's last block first)
    const int Sq = p.Sq, Sk = p.Sk;
    const int q0w = w.blk * kQB + wave * 32;
    const int qrow = q0w + l31;

    const float* __restrict__ kg = p.k + (size_t)(w.b * p.Hkv + w.hk) * Sk * D;
    const float* __restrict__ vg = p.v + (size_t)(w.b * p.Hkv + w.hk) * Sk * D;

    // Q (B operand): lane (q, hi) step s holds Q[q][2s + hi]
    float qf[D / 2];
    {
        const int qr = qrow < Sq ? qrow : Sq - 1;
        const float* qp = p.q + ((size_t)(w.b * p.Hq + w.h) * Sq + qr) * D;
#pragma unroll
        for (int s4 = 0; s4 < D / 4; ++s4) {
            const f32x4_t x = *reinterpret_cast<const f32x4_t*>(qp + 4 * s4);
            qf[2 * s4] = hi ? x[1] : x[0];
            qf[2 * s4 + 1] = hi ? x[3] : x[2];
        }
    }

    f32x16_t o[DB];
#pragma unroll
    for (int d = 0; d < DB; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
    float m = -INFINITY, l = 0.f;
    const float c = p.c;

    const int coff = p.coff;
    const int kv_hi = CAUSAL ? min(Sk, w.blk * kQB + kQB + coff) : Sk;
    const int nt = (kv_hi + kKV - 1) / kKV;
    const int wave_kv_hi = CAUSAL ? min(Sk, q0w + 32 + coff) : Sk;
    const int W = p.window;
    int t_lo = W > 0 ? max(0, w.blk * kQB + coff - W + 1) / kKV : 0;   // tiles before the block's window: skipped
    const int wave_kv_lo = W > 0 ? q0w + coff - W + 1 : 0;                  // first key any row of this wave can see
    int nt_end = nt;
    if (p.npiece > 1) {   // this piece's share of the block's tiles
        const int chunk = (max(0, nt - t_lo) + p.npiece - 1) / p.npiece;
        t_lo = t_lo + piece * chunk;
        nt_end = min(nt, t_lo + chunk);
    }

    // Staging is software-pipelined: the next tile's K / V rows are requested into registers before this tile's MFMAs and
    // written to LDS behind the barrier that retires this tile (the loads used to sit, latency exposed, at the top of
    // every iteration of a single-buffered loop).
    f32x4_t kx[NCH], vx[NCH];
    auto issue_tile = [&](int t) __attribute__((always_inline)) {
        const int kv0 = t * kKV;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int cidx = tid + 256 * i;
            {   // K: consecutive lanes -> consecutive kv rows, same d chunk
                const int kv = cidx & 31, dc = cidx >> 5;
                int r = kv0 + kv;
                r = r < Sk ? r : Sk - 1;
                kx[i] = *reinterpret_cast<const f32x4_t*>(kg + (size_t)r * D + 4 * dc);
            }
            {   // V: coalesced rows
                const int row = cidx / C4, cc = cidx % C4;
                int r = kv0 + row;
                r = r < Sk ? r : Sk - 1;
                vx[i] = *reinterpret_cast<const f32x4_t*>(vg + (size_t)r * D + 4 * cc);
            }
        }
    };
    auto write_tile = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int cidx = tid + 256 * i;
            const int kv = cidx & 31, dc = cidx >> 5;
            Kt[(4 * dc + 0) * 32 + kv] = kx[i][0];
            Kt[(4 * dc + 1) * 32 + kv] = kx[i][1];
            Kt[(4 * dc + 2) * 32 + kv] = kx[i][2];
            Kt[(4 * dc + 3) * 32 + kv] = kx[i][3];
            const int row = cidx / C4, cc = cidx % C4;
            *reinterpret_cast<f32x4_t*>(&Vs[row * D + 4 * cc]) = vx[i];
        }
    };
    if (t_lo < nt_end) issue_tile(t_lo);
    for (int t = t_lo; t < nt_end; ++t) {
        const int kv0 = t * kKV;
        write_tile();
        __syncthreads();
        if (t + 1 < nt_end) issue_tile(t + 1);

        if (kv0 < wave_kv_hi && kv0 + kKV > wave_kv_lo) {
            f32x16_t s;
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
            for (int st = 0; st < D / 2; ++st)
                s = __builtin_amdgcn_mfma_f32_32x32x2f32(Kt[(2 * st + hi) * 32 + l31], qf[st], s, 0, 0, 0);

            // t = c * s ; mask ; online softmax in the exp2 domain
            const bool need_mask = (CAUSAL && (kv0 + kKV - 1 > q0w + coff)) || (kv0 + kKV > Sk) || (W > 0 && q0w + coff + 31 - kv0 >= W);
            float mx = -INFINITY;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float x = s[r] * c;
                if (need_mask) {
                    const int kv = kv0 + crow(r, hi);
                    const bool vis = (kv < Sk) && (!CAUSAL || kv <= qrow + coff) && (W <= 0 || qrow + coff - kv < W);
                    x = vis ? x : -INFINITY;
                }
                s[r] = x;
                mx = fmaxf(mx, x);
            }
            mx = fmaxf(mx, xhalf(mx));
            const float m_new = fmaxf(m, mx);
            const float m_use = (m_new == -INFINITY) ? 0.f : m_new;  // row without a visible key so far (window)
            const float alpha = (m_new == -INFINITY) ? 1.0f : fast_exp2(m - m_new);
            m = m_new;
            l *= alpha;
#pragma unroll
            for (int d = 0; d < DB; ++d)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[d][r] *= alpha;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                s[r] = fast_exp2(s[r] - m_use);
                l += s[r];
            }
            // O^T += V^T . P^T : step r contracts kv pair {crow(r,0), crow(r,1)}
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int kvr = (r & 3) + 8 * (r >> 2) + 4 * hi;
#pragma unroll
                for (int d = 0; d < DB; ++d)
                    o[d] = __builtin_amdgcn_mfma_f32_32x32x2f32(Vs[kvr * D + 32 * d + l31], s[r], o[d], 0, 0, 0);
            }
        }
        __syncthreads();
    }

    const float lt = l + xhalf(l);
    if (p.npiece > 1) {   // partial row: un-normalised O, m, l
        if (qrow < Sq) {
            float* prow = p.part + ((size_t)piece * p.rows + (size_t)(w.b * p.Hq + w.h) * Sq + qrow) * (D + 4);
#pragma unroll
            for (int d = 0; d < DB; ++d)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    f32x4_t x = {o[d][4 * g4], o[d][4 * g4 + 1], o[d][4 * g4 + 2], o[d][4 * g4 + 3]};
                    *reinterpret_cast<f32x4_t*>(prow + 32 * d + 8 * g4 + 4 * hi) = x;
                }
            if (hi == 0) {
                prow[D] = m;
                prow[D + 1] = lt;
            }
        }
        return;
    }
    const float inv = lt > 0.f ? 1.0f / lt : 0.f;  // no visible key at all (window beyond Sk): O = 0, LSE = -inf
    if (qrow < Sq) {
        float* orow = p.o + ((size_t)(w.b * p.Hq + w.h) * Sq + qrow) * D;
#pragma unroll
        for (int d = 0; d < DB; ++d)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                f32x4_t x = {o[d][4 * g4] * inv, o[d][4 * g4 + 1] * inv, o[d][4 * g4 + 2] * inv,
                             o[d][4 * g4 + 3] * inv};
                *reinterpret_cast<f32x4_t*>(orow + 32 * d + 8 * g4 + 4 * hi) = x;
            }
        if (p.lse != nullptr && hi == 0)
            p.lse[(size_t)(w.b * p.Hq + w.h) * Sq + qrow] = (m + fast_log2(lt)) * kLn2;
    }
}

// merge of the key-range pieces: O = sum_j w_j O_j / sum_j w_j l_j, w_j = 2^(m_j - max m); D / 4 threads per row
struct CombineF32Params {
    const float* part;
    float* o;
    float* lse;
    long long rows;
    int npiece;
};

template <int D>
__global__ void __launch_bounds__(256) fa_fwd_f32_combine(const CombineF32Params p) {
    constexpr int TPR = D / 4, RPB = 256 / TPR;
    const long long row = (long long)blockIdx.x * RPB + threadIdx.x / TPR;
    const int c = threadIdx.x % TPR;
    if (row >= p.rows) return;
    float mmax = -INFINITY;
    for (int j = 0; j < p.npiece; ++j) mmax = fmaxf(mmax, p.part[((size_t)j * p.rows + row) * (D + 4) + D]);
    f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
    float l = 0.f;
    for (int j = 0; j < p.npiece; ++j) {
        const float* pr = p.part + ((size_t)j * p.rows + row) * (D + 4);
        const float mj = pr[D];
        const float wj = (mj == -INFINITY) ? 0.f : fast_exp2(mj - mmax);
        const f32x4_t x = *reinterpret_cast<const f32x4_t*>(pr + 4 * c);
        acc[0] += wj * x[0]; acc[1] += wj * x[1]; acc[2] += wj * x[2]; acc[3] += wj * x[3];
        l += wj * pr[D + 1];
    }
    const float inv = l > 0.f ? 1.0f / l : 0.f;
    f32x4_t y = {acc[0] * inv, acc[1] * inv, acc[2] * inv, acc[3] * inv};
    *reinterpret_cast<f32x4_t*>(p.o + row * D + 4 * c) = y;
    if (p.lse != nullptr && c == 0) p.lse[row] = (mmax + fast_log2(l)) * kLn2;
}

// Pieces per Q block for grids that leave most of the chip idle (the reference's own Zig benchmark shape, tests/benchmark_attention.zig:
// 18-21: B4 H8 S512 D64 = 128 workgroups for 512 slots): as many as fill the slots, at least two key tiles each, at most 8.
// AULE_HIP_F32_SPLIT=0 turns it off (A/B).
inline int f32_pieces(const FwdArgs& a, int nqb) {
    static const int on = [] {
        const char* e = std::getenv("AULE_HIP_F32_SPLIT");
        return (e != nullptr && e[0] == '0') ? 0 : 1;
    }();
    if (!on) return 1;
    const long long items = (long long)nqb * a.B * a.Hq, slots = 2LL * device_cu_count(a.device);
    if (items <= 0 || items * 2 > slots) return 1;
    const int kv_hi = a.causal ? (a.Sk < a.Sq + a.coff ? a.Sk : a.Sq + a.coff) : a.Sk;   // the largest block's keys
    const int tiles = (kv_hi + kKV - 1) / kKV;
    long long n = slots / items;
    if (n > tiles / 2) n = tiles / 2;
    if (n > 8) n = 8;
    return n < 2 ? 1 : (int)n;
}

template <int D>
int launch_f32(const FwdArgs& a, hipStream_t stream) {
    FwdF32Params p;
    p.q = (const float*)a.q; p.k = (const float*)a.k; p.v = (const float*)a.v;
    p.o = (float*)a.o; p.lse = a.lse;
    p.B = a.B; p.Hq = a.Hq; p.Hkv = a.Hkv; p.Sq = a.Sq; p.Sk = a.Sk;
    p.c = a.scale * kLog2e;
    p.nqb = (a.Sq + kQB - 1) / kQB;
    p.window = a.window > 0 ? a.window : 0;
    p.coff = a.causal ? a.coff : 0;
    p.npiece = f32_pieces(a, p.nqb);
    p.part = nullptr;
    p.rows = (long long)a.B * a.Hq * a.Sq;
    const uint64_t bytes = p.npiece > 1 ? (uint64_t)p.npiece * p.rows * (D + 4) * sizeof(float) : 0;
    if (a.query_ws != nullptr) {
        *a.query_ws = bytes;
        return 0;
    }
    const dim3 grid((unsigned)(p.nqb * a.B * a.Hq * p.npiece)), block(256);
    if (p.npiece == 1) {
        if (a.causal)
            hipLaunchKernelGGL((fa_fwd_f32_kernel<D, true>), grid, block, 0, stream, p);
        else
            hipLaunchKernelGGL((fa_fwd_f32_kernel<D, false>), grid, block, 0, stream, p);
        return (int)hipGetLastError();
    }
    ScopedWorkspace ws(bytes, a.ws, a.ws_bytes, stream);
    if (ws.err != hipSuccess) return (int)ws.err;
    p.part = static_cast<float*>(ws.ptr);
    if (a.causal)
        hipLaunchKernelGGL((fa_fwd_f32_kernel<D, true>), grid, block, 0, stream, p);
    else
        hipLaunchKernelGGL((fa_fwd_f32_kernel<D, false>), grid, block, 0, stream, p);
    int rc = (int)hipGetLastError();
    if (rc) return rc;
    CombineF32Params c;
    c.part = p.part; c.o = p.o; c.lse = p.lse; c.rows = p.rows; c.npiece = p.npiece;
    constexpr int RPB = 256 / (D / 4);
    hipLaunchKernelGGL((fa_fwd_f32_combine<D>), dim3((unsigned)((p.rows + RPB - 1) / RPB)), dim3(256), 0, stream, c);
    return (int)hipGetLastError();
}

}  // namespace

int launch_fwd_f32(const FwdArgs& a, hipStream_t stream) {
    if (a.D == 128) return launch_f32<128>(a, stream);
    if (a.D == 64) return launch_f32<64>(a, stream);
    if (a.D == 32) return launch_f32<32>(a, stream);
    return -1;
}

int configure_fwd_f32() { return 0; }

void work_order_dump(int ranked, int bid, int B, int Hq, int Hkv, int nblk, int flag, int* out4) {
    const WorkItem w = ranked ? decode_work_ranked(bid, B, Hq, Hkv, nblk, flag != 0) : decode_work(bid, B, Hq, Hkv, nblk, flag != 0);
    out4[0] = w.b; out4[1] = w.hk; out4[2] = w.h; out4[3] = w.blk;
}

}  // namespace aule_hip
