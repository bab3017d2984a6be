// fa_fwd_gfx950.hip -- FlashAttention-2 forward for MI355X (gfx950 / CDNA4).
//
// Replaces, behind aule_attention_forward_ex, the Triton launch of the reference
// (python/aule/triton_flash_amd.py:393-445 FlashAttentionAMDFunc.forward and its
// kernel :97-240) and the HIP stub slot (src/backends/attention_hip.cpp:22-119).
// Numerics contract: s = scale * q.k ; top-left causal
// mask (j <= i) ; online softmax with fp32 m, l, acc ; P cast to the V dtype
// before the PV product (triton_flash_amd.py:222) ; LSE = m + ln(l) (:237).
//
// Design (16-bit kernels; this is not a translation of the Triton kernel):
//   * workgroup = 8 wavefronts x 32 query rows = one 256-row Q block; KV tiles of
//     64 rows, double-buffered in LDS, staged global->VGPR->LDS with the loads
//     issued one tile ahead (their HBM/L2 latency hides under the MFMAs).
//   * "swapped" QK^T: S^T[kv][q] = K_tile . Q^T so that every lane owns ONE query
//     row (q = lane & 31): row max / row sum / rescale factors are lane-local,
//     the only cross-lane traffic is one exchange between the two 32-lane halves.
//   * The fp32 S^T accumulator layout (kv = (r&3)+8(r>>2)+4*half) is used
//     directly as the k-slot order of the PV MFMA: P never leaves registers and
//     V^T fragments are fetched with ds_read_b64_tr_b16 in that same kv order.
//   * O^T[d][q] = V^T . P^T is accumulated so the per-row rescale is again
//     lane-local.
//   * K tile in LDS is row-major with a 16-byte-chunk XOR swizzle (conflict-free
//     ds_read_b128 for the MFMA A operand); V tile is stored as [kv/4][d/16][4][16]
//     sub-tiles, the layout the transpose-read gathers without bank conflicts.
//   * grid order: heaviest causal Q blocks first; all blocks of one (batch,
//     kv-head) on one XCD so K/V are fetched from HBM once per XCD L2.
#include <cstdlib>

#include "fa_device.h"
#include "fa_kernels.h"

namespace aule_hip {
namespace {

struct FwdParams {
    const void* q;
    const void* k;
    const void* v;
    void* o;
    float* lse;
    int B, Hq, Hkv, Sq, Sk;
    float c;     // |scale| * log2(e)
    int negq;    // scale < 0: flip the sign of Q on load
    int nqb;     // number of 256-row Q blocks
};

constexpr int kQBlock = 256;  // query rows per workgroup (8 waves x 32)
constexpr int kKVTile = 64;

template <int D>
struct FwdCfg {
    static constexpr int RB = D * 2;                  // bytes per K/V row
    static constexpr int CPR = RB / 16;               // 16-byte chunks per row
    static constexpr int TILE = kKVTile * RB;         // bytes per K (or V) tile
    static constexpr int NCHUNK = TILE / 16;          // 16-byte chunks per tile
    static constexpr int CH = (NCHUNK + 511) / 512;   // chunks per thread per tile
    static constexpr int KS = D / 16;                 // k-steps of QK^T
    static constexpr int DB = D / 32;                 // 32-wide d blocks of O
    static constexpr int LDS = 4 * TILE;              // K x2 + V x2
};

// K swizzle: physical 16-B chunk = chunk ^ swz(row); makes the 16 rows touched by
// one ds_read_b128 lane group land on 16 distinct 16-B slots of the 256-B bank row.
template <int D>
__device__ __forceinline__ int kswz(int row) {
    if constexpr (D == 128) return row & 15;
    else if constexpr (D == 64) return (row >> 1) & 7;
    else return (row >> 2) & 3;  // D == 32: 64-B rows, 4 rows per bank row
}

template <class T, int D, bool CAUSAL>
__global__ void __launch_bounds__(512) fa_fwd_kernel(const FwdParams p) {
    using Cfg = FwdCfg<D>;
    using v8 = typename T::v8;
    constexpr int RB = Cfg::RB, CPR = Cfg::CPR, TILE = Cfg::TILE, CH = Cfg::CH, KS = Cfg::KS, DB = Cfg::DB;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const Ks = smem;
    char* const Vs = smem + 2 * TILE;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31;
    const int hi = lane >> 5;

    const WorkItem w = decode_work(blockIdx.x, p.B, p.Hq, p.Hkv, p.nqb, CAUSAL);
    const int Sq = p.Sq, Sk = p.Sk;
    const int q0w = w.blk * kQBlock + wave * 32;  // first query row of this wave
    const int qrow = q0w + l31;

    const u32x4_t* __restrict__ kg =
        reinterpret_cast<const u32x4_t*>(p.k) + (size_t)(w.b * p.Hkv + w.hk) * Sk * CPR;
    const u32x4_t* __restrict__ vg =
        reinterpret_cast<const u32x4_t*>(p.v) + (size_t)(w.b * p.Hkv + w.hk) * Sk * CPR;

    // ---- Q fragments (B operand of S^T = K.Q^T): lane (q, hi) holds d = 16ks+8hi..+7
    v8 qf[KS];
    {
        const int qr = qrow < Sq ? qrow : Sq - 1;
        const u32x4_t* qp =
            reinterpret_cast<const u32x4_t*>(p.q) + ((size_t)(w.b * p.Hq + w.h) * Sq + qr) * CPR;
        const unsigned flip = p.negq ? 0x80008000u : 0u;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            u32x4_t x = qp[2 * ks + hi];
            x[0] ^= flip; x[1] ^= flip; x[2] ^= flip; x[3] ^= flip;
            qf[ks] = as_v8<T>(x);
        }
    }

    // ---- staging map: thread -> CH 16-byte chunks of the 64 x D tile
    int st_row[CH], st_goff[CH], st_koff[CH], st_voff[CH];
#pragma unroll
    for (int i = 0; i < CH; ++i) {
        const int c = tid + 512 * i;
        const int row = c / CPR, cc = c % CPR;
        st_row[i] = row;
        st_goff[i] = cc;
        st_koff[i] = row * RB + ((cc ^ kswz<D>(row)) << 4);
        st_voff[i] = ((row >> 2) * (D / 16) + (cc >> 1)) * 128 + (row & 3) * 32 + (cc & 1) * 16;
    }

    // ---- per-lane LDS read offsets
    int ka_off[KS];  // K A-operand: row l31 (+32*sb), chunk 2ks+hi, swizzled
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) ka_off[ks] = l31 * RB + (((2 * ks + hi) ^ kswz<D>(l31)) << 4);
    // for sb = 1 the row is l31 + 32: kswz must agree -> true for D=128 (row&15) and
    // D=64 ((row>>1)&7: 32>>1 = 16 = 0 mod 8) and D=32 ((row>>2)&3: 8 = 0 mod 4).
    // V^T A-operand via transpose read: see fa_device.h and DESIGN.md
    const int va_off = hi * (D / 16) * 128 + ((lane >> 4) & 1) * 128 + (lane & 15) * 8;

    const int kv_hi = CAUSAL ? min(Sk, w.blk * kQBlock + kQBlock) : Sk;
    const int nt = (kv_hi + kKVTile - 1) / kKVTile;
    const int wave_kv_hi = CAUSAL ? min(Sk, q0w + 32) : Sk;  // keys visible to some row of this wave

    u32x4_t kst[CH], vst[CH];
    auto issue_loads = [&](int kv0) {
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            if (Cfg::NCHUNK % 512 == 0 || tid + 512 * i < Cfg::NCHUNK) {
                int r = kv0 + st_row[i];
                r = r < Sk ? r : Sk - 1;
                kst[i] = kg[(size_t)r * CPR + st_goff[i]];
                vst[i] = vg[(size_t)r * CPR + st_goff[i]];
            }
        }
    };
    auto write_stage = [&](int buf) {
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            if (Cfg::NCHUNK % 512 == 0 || tid + 512 * i < Cfg::NCHUNK) {
                *reinterpret_cast<u32x4_t*>(Ks + buf * TILE + st_koff[i]) = kst[i];
                *reinterpret_cast<u32x4_t*>(Vs + buf * TILE + st_voff[i]) = vst[i];
            }
        }
    };

    f32x16_t o[DB];
#pragma unroll
    for (int d = 0; d < DB; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
    float m = -INFINITY;  // running max, log2 domain (scaled by c)
    float l = 0.f;        // running sum (this half-wave's share)
    const float c = p.c;

    issue_loads(0);
    write_stage(0);
    __syncthreads();

    for (int t = 0; t < nt; ++t) {
        const int cur = t & 1;
        const int kv0 = t * kKVTile;
        if (t + 1 < nt) issue_loads(kv0 + kKVTile);

        if (kv0 < wave_kv_hi) {
            const char* kb = Ks + cur * TILE;
            const char* vb = Vs + cur * TILE + va_off;

            // ---- S^T = K . Q^T  (2 x [32 kv x 32 q], fp32)
            f32x16_t s[2];
#pragma unroll
            for (int sb = 0; sb < 2; ++sb)
#pragma unroll
                for (int r = 0; r < 16; ++r) s[sb][r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
                for (int sb = 0; sb < 2; ++sb) {
                    const u32x4_t a = *reinterpret_cast<const u32x4_t*>(kb + ka_off[ks] + sb * 32 * RB);
                    s[sb] = T::mfma(as_v8<T>(a), qf[ks], s[sb]);
                }
            }

            // ---- masks (diagonal tiles and the ragged last tile only)
            const bool need_mask = (CAUSAL && (kv0 + kKVTile - 1 > q0w)) || (kv0 + kKVTile > Sk);
            if (need_mask) {
#pragma unroll
                for (int sb = 0; sb < 2; ++sb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int kv = kv0 + sb * 32 + crow(r, hi);
                        const bool vis = (kv < Sk) && (!CAUSAL || kv <= qrow);
                        s[sb][r] = vis ? s[sb][r] : -INFINITY;
                    }
            }

            // ---- online softmax (exp2 domain), all lane-local except one half exchange
            float mx = s[0][0];
#pragma unroll
            for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[0][r]);
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[1][r]);
            mx = fmaxf(mx, xhalf(mx));
            const float m_new = fmaxf(m, mx * c);
            const float alpha = fast_exp2(m - m_new);
            m = m_new;
            l *= alpha;
#pragma unroll
            for (int d = 0; d < DB; ++d)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[d][r] *= alpha;

            v8 pb[2][2];
#pragma unroll
            for (int sb = 0; sb < 2; ++sb) {
                float pr[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    pr[r] = fast_exp2(__builtin_fmaf(s[sb][r], c, -m_new));
                    l += pr[r];
                }
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    u32x4_t u;
#pragma unroll
                    for (int j = 0; j < 4; ++j) u[j] = T::pack2(pr[8 * kk + 2 * j], pr[8 * kk + 2 * j + 1]);
                    pb[sb][kk] = as_v8<T>(u);
                }
            }

            // ---- O^T += V^T . P^T   (A = V^T via LDS transpose read, B = P in registers)
#pragma unroll
            for (int sb = 0; sb < 2; ++sb)
#pragma unroll
                for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                    for (int d = 0; d < DB; ++d) {
                        const int off = ((8 * sb + 4 * kk) * (D / 16) + 2 * d) * 128;
                        const s16x4_t a0 = lds_tr16(vb + off);
                        const s16x4_t a1 = lds_tr16(vb + off + 2 * (D / 16) * 128);
                        o[d] = T::mfma(as_v8<T>(a0, a1), pb[sb][kk], o[d]);
                    }
        }

        if (t + 1 < nt) write_stage(cur ^ 1);
        __syncthreads();
    }

    // ---- epilogue: O = O^T / l ; LSE = (m + log2 l) * ln2
    const float lt = l + xhalf(l);
    const float inv = 1.0f / lt;
    if (qrow < Sq) {
        char* orow = reinterpret_cast<char*>(p.o) + ((size_t)(w.b * p.Hq + w.h) * Sq + qrow) * RB;
#pragma unroll
        for (int d = 0; d < DB; ++d)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                u32x2_t u;
                u[0] = T::pack2(o[d][4 * g4 + 0] * inv, o[d][4 * g4 + 1] * inv);
                u[1] = T::pack2(o[d][4 * g4 + 2] * inv, o[d][4 * g4 + 3] * inv);
                *reinterpret_cast<u32x2_t*>(orow + (32 * d + 8 * g4 + 4 * hi) * 2) = u;
            }
        if (p.lse != nullptr && hi == 0)
            p.lse[(size_t)(w.b * p.Hq + w.h) * Sq + qrow] = (m + fast_log2(lt)) * kLn2;
    }
}

template <class T, int D>
int launch_fwd_16(const FwdArgs& a, hipStream_t stream) {
    FwdParams p;
    p.q = a.q; p.k = a.k; p.v = a.v; p.o = a.o; p.lse = a.lse;
    p.B = a.B; p.Hq = a.Hq; p.Hkv = a.Hkv; p.Sq = a.Sq; p.Sk = a.Sk;
    float c = a.scale * kLog2e;
    p.negq = c < 0.f;
    c = c < 0.f ? -c : c;
    if (c == 0.f) c = 1e-30f;  // scale == 0: uniform weights over the visible keys
    p.c = c;
    p.nqb = (a.Sq + kQBlock - 1) / kQBlock;
    const dim3 grid((unsigned)(p.nqb * a.B * a.Hq));
    const dim3 block(512);
    const size_t lds = FwdCfg<D>::LDS;
    if (a.causal)
        hipLaunchKernelGGL((fa_fwd_kernel<T, D, true>), grid, block, lds, stream, p);
    else
        hipLaunchKernelGGL((fa_fwd_kernel<T, D, false>), grid, block, lds, stream, p);
    return (int)hipGetLastError();
}

template <class T, int D>
int set_attr_16() {
    int rc = (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&fa_fwd_kernel<T, D, true>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, FwdCfg<D>::LDS);
    if (rc) return rc;
    return (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&fa_fwd_kernel<T, D, false>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, FwdCfg<D>::LDS);
}

}  // namespace

int launch_fwd_f32(const FwdArgs& a, hipStream_t stream);  // fa_fwd_f32.hip
int configure_fwd_f32();
int launch_fwd_pp(const FwdArgs& a, hipStream_t stream);   // fa_fwd_pp_gfx950.hip
int launch_fwd_pp_split(const FwdArgs& a, hipStream_t stream);
bool pp_split_applicable(const FwdArgs& a);
int configure_fwd_pp();
int launch_fwd_iw(const FwdArgs& a, hipStream_t stream);   // fa_fwd_iw_gfx950.hip (-1: shape not covered)
int configure_fwd_iw();
int launch_fwd_ps(const FwdArgs& a, hipStream_t stream);   // fa_fwd_ps_gfx950.hip (persistent tile stream)
bool fwd_ps_applicable(const FwdArgs& a);
int configure_fwd_ps();

// AULE_HIP_FWD_KERNEL = "pp" (8-wave ping-pong schedule) | "iw" (4-wave in-wave ping-pong, D = 128) |
// "v1" (one barrier per tile, all waves in the same phase); the non-default ones are kept for A/B measurements
static int fwd_kernel_choice() {
    static const int v = [] {
        const char* e = getenv("AULE_HIP_FWD_KERNEL");
        if (e != nullptr && e[0] == 'v' && e[1] == '1') return 1;
        if (e != nullptr && e[0] == 'i' && e[1] == 'w') return 2;
        if (e != nullptr && e[0] == 'p' && e[1] == 'p') return 4;   // one workgroup per Q-block pair (the stream's predecessor)
        return 0;
    }();
    return v;
}
static bool use_v1() { return fwd_kernel_choice() == 1; }
static bool use_ps(const FwdArgs& a) { return fwd_kernel_choice() == 0 && fwd_ps_applicable(a); }

bool splitkv_applicable(const FwdArgs& a);                      // fa_fwd_splitkv_gfx950.hip
int launch_fwd_splitkv(const FwdArgs& a, hipStream_t stream);

// AULE_HIP_FWD_SPLITKV=0 keeps short-query shapes on the tiled kernels (A/B measurements)
static bool splitkv_enabled() {
    static const int v = [] {
        const char* e = getenv("AULE_HIP_FWD_SPLITKV");
        return (e != nullptr && e[0] == '0') ? 0 : 1;
    }();
    return v == 1;
}

// Non-causal problems that the plain tiled launch would run badly (few workgroups, or Q blocks mostly without rows):
// 4 = wave-per-chunk split-KV kernel, 5 = tiled kernel with packed rows + KV splits, 0 = neither.  Measured on one
// box per comparison (tools/ppsplit_grid.py, tools/ppsplit_decode.py): the tiled variant wins almost
// everywhere, including Sq = 1 (its combine merges <= 32 partials per row, the wave kernel's hundreds); the wave kernel
// keeps the pure streaming corner -- many units, at most half a row tile of packed rows, K+V beyond ~100 MB -- where
// it is 10-15 % ahead at D = 128 and 35-55 % at D = 64.  Differences below ~8 % on these kernels are noise.
static int short_query_route(const FwdArgs& a) {
    if (a.dtype == kF32 || a.window > 0) return 0;
    const bool wave_ok = !a.causal && splitkv_enabled() && splitkv_applicable(a);   // (the wave kernel has no mask)
    const bool tiled_ok = pp_split_applicable(a) && !use_v1() && fwd_kernel_choice() != 2;
    if (wave_ok && tiled_ok) {
        const long long units = (long long)a.B * a.Hkv;
        const long long rows = (long long)(a.Hq / a.Hkv) * a.Sq;
        const double kv_bytes = 2.0 * (double)units * a.Sk * a.D * 2.0;
        return (units >= 32 && rows <= 16 && kv_bytes >= 100e6) ? 4 : 5;
    }
    return wave_ok ? 4 : (tiled_ok ? 5 : 0);
}

// Which kernel launch_fwd() picks for `a` (host logic only, no device work): 0 fp32, 1 ping-pong, 2 in-wave,
// 3 lock-step v1, 4 split-KV, 5 tiled kernel with packed rows + KV splits.  Lets the tests pin the path a shape exercises (the in-wave kernel can still decline
// a shape at launch and fall through to the ping-pong kernel).
int fwd_route(const FwdArgs& a) {
    if (a.dtype == kF32) return 0;
    const int sq = short_query_route(a);
    if (sq) return sq;
    if (use_ps(a)) return 6;
    const bool pp_only = a.window > 0 || (a.causal && a.coff != 0);
    if (fwd_kernel_choice() == 2 && !pp_only) return 2;
    if (!use_v1() || pp_only) return 1;
    return 3;
}

uint64_t fwd_workspace_bytes(FwdArgs a) {
    uint64_t bytes = 0;
    a.query_ws = &bytes;
    (void)launch_fwd(a, nullptr);   // dry run: the two-launch paths report their plan, the others launch nothing
    return bytes;
}

uint64_t paged_workspace_bytes(PagedArgs a) {
    uint64_t bytes = 0;
    a.query_ws = &bytes;
    (void)launch_paged_decode(a, nullptr);
    return bytes;
}

int launch_fwd(const FwdArgs& a, hipStream_t stream) {
    if (a.query_ws != nullptr) *a.query_ws = 0;
    const int sq = a.dtype == kF32 ? 0 : short_query_route(a);
    if (sq == 4) return launch_fwd_splitkv(a, stream);
    if (sq == 5) return launch_fwd_pp_split(a, stream);
    if (a.query_ws != nullptr) return 0;   // single-launch paths need no workspace
    if (a.dtype == kF32) return launch_fwd_f32(a, stream);
    if (use_ps(a)) return launch_fwd_ps(a, stream);
    const bool pp_only = a.window > 0 || (a.causal && a.coff != 0);  // window / shifted causal live in the ping-pong kernel
    if (fwd_kernel_choice() == 2 && !pp_only) {
        const int rc = launch_fwd_iw(a, stream);
        if (rc != -1) return rc;
    }
    if (!use_v1() || pp_only) return launch_fwd_pp(a, stream);
    if (a.dtype == kBF16) {
        if (a.D == 128) return launch_fwd_16<Bf16Traits, 128>(a, stream);
        if (a.D == 64) return launch_fwd_16<Bf16Traits, 64>(a, stream);
        if (a.D == 32) return launch_fwd_16<Bf16Traits, 32>(a, stream);
    } else if (a.dtype == kF16) {
        if (a.D == 128) return launch_fwd_16<F16Traits, 128>(a, stream);
        if (a.D == 64) return launch_fwd_16<F16Traits, 64>(a, stream);
        if (a.D == 32) return launch_fwd_16<F16Traits, 32>(a, stream);
    }
    return -1;
}

int configure_fwd() {
    int rc = 0;
    rc |= set_attr_16<Bf16Traits, 128>();
    rc |= set_attr_16<Bf16Traits, 64>();
    rc |= set_attr_16<Bf16Traits, 32>();
    rc |= set_attr_16<F16Traits, 128>();
    rc |= set_attr_16<F16Traits, 64>();
    rc |= set_attr_16<F16Traits, 32>();
    rc |= configure_fwd_f32();
    rc |= configure_fwd_pp();
    rc |= configure_fwd_ps();
    rc |= configure_fwd_iw();
    return rc;
}

}  // namespace aule_hip
