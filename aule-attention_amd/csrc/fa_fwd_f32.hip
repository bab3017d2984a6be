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

    const WorkItem w = decode_work_ranked(blockIdx.x, p.B, p.Hq, p.Hkv, p.nqb, CAUSAL);   // (causal: every unit's last block first)
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
    const int t_lo = W > 0 ? max(0, w.blk * kQB + coff - W + 1) / kKV : 0;   // tiles before the block's window: skipped
    const int wave_kv_lo = W > 0 ? q0w + coff - W + 1 : 0;                  // first key any row of this wave can see

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
    if (t_lo < nt) issue_tile(t_lo);
    for (int t = t_lo; t < nt; ++t) {
        const int kv0 = t * kKV;
        write_tile();
        __syncthreads();
        if (t + 1 < nt) issue_tile(t + 1);

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
    const dim3 grid((unsigned)(p.nqb * a.B * a.Hq)), block(256);
    if (a.causal)
        hipLaunchKernelGGL((fa_fwd_f32_kernel<D, true>), grid, block, 0, stream, p);
    else
        hipLaunchKernelGGL((fa_fwd_f32_kernel<D, false>), grid, block, 0, stream, p);
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
