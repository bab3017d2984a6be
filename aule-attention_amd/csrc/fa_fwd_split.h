// fa_fwd_split.h -- small grids: every pair of causal Q blocks (or every non-causal block) cut into pieces along the KEYS.
//
// Single-sequence prefill (B1 H8 S8192: 128 paired items on 256 CUs) leaves most of the chip idle, and a paired causal launch
// cannot be made finer by cutting query rows: a workgroup's time is its key tile steps.  So a pair becomes n work items of 1/n of
// its key tiles; a block that was cut leaves one fp32 partial row per query and piece (un-normalised O, the softmax reference in
// log2 units, the row sum) and one more launch merges them.  The forward kernel walks a list of parts anyway: the cut is a change
// of the list (first tile and tile count per part), not of the tile loop.  (DESIGN.md 3.2c.  Round 2 built this on the
// two-waves-per-SIMD stream kernel; round 4 moved it to the one-wave-per-SIMD kernel and retired that file.)
//
// Shared by the host plan, the forward kernel's part table and the merge kernel; pinned on CPU through
// aule_hip_debug_forward_split_plan (tests/test_capi_symbols.py).  Replaces nothing in the reference (its launch grid is one
// program per 128-row block whatever the problem: python/aule/triton_flash_amd.py:434-445).
#pragma once
#include "fa_device.h"
#include "fa_kernels.h"
#include "fa_fwd_tile.h"

namespace aule_hip {
namespace {

constexpr int kPartPad = 4;   // floats behind the D accumulators of a partial row: reference, row sum, 2 unused (rows stay 16-byte aligned)
constexpr int kMaxPieces = 8;
constexpr int kEverything = 1 << 30;   // position offset of a non-causal problem (every key visible to every row)
constexpr int kSplitMinTiles = 16;     // shortest piece worth a workgroup of its own (split_plan)

// KV tiles of Q block qb under the causal rule (query i at position i + coff).
__host__ __device__ inline int split_tiles(int qb, int Sk, int coff) {
    int kv_hi = qb * kQBlock + kQBlock + coff;
    kv_hi = kv_hi < Sk ? kv_hi : Sk;
    kv_hi = kv_hi > 1 ? kv_hi : 1;
    return (kv_hi + kKVTile - 1) / kKVTile;
}
// Plan of the pair (far, near) of Q blocks: the pair's key tiles, far block's first, are one sequence of ntf + ntn tiles cut into
// n pieces of (nearly) equal length, piece j = [b[j], b[j + 1]) -- at most one range of each block.  A cut inside a block stays
// within the keys EVERY row of the block sees whole (tile index <= first position / 64): ranges in front of it need no mask, and
// every row of the range behind it sees that range's first key, so every range runs the ordinary softmax (a finite reference
// from its first tile).  Every range has at least four tiles (the part prologue consumes two and requests a third); a cut with no
// admissible position collapses (b[j] = b[j - 1]: an empty piece).
struct SplitPair {
    int ntf, ntn;
    int b[kMaxPieces + 1];
};
__host__ __device__ inline unsigned split_magic(int n) { return (unsigned)((0x100000000ull + (unsigned)n - 1) / (unsigned)n); }
// (magic = split_magic(n), computed on the host: T / n as a multiply-high, exact for T < 2^29 -- a division would drag the whole
// plan from the scalar unit into vector registers)
__host__ __device__ inline SplitPair split_cuts(int far, int near, int Sk, int coff, int n, unsigned magic) {
    SplitPair r;
    r.ntf = split_tiles(far, Sk, coff);
    r.ntn = far != near ? split_tiles(near, Sk, coff) : 0;
    const int T = r.ntf + r.ntn;
    int fhi = (far * kQBlock + coff) / kKVTile, nhi = (near * kQBlock + coff) / kKVTile;   // last admissible cut inside a block
    fhi = fhi < r.ntf - 4 ? fhi : r.ntf - 4;
    nhi = nhi < r.ntn - 4 ? nhi : r.ntn - 4;
    r.b[0] = 0;
    const int q = (int)(((unsigned long long)(unsigned)T * magic) >> 32), rem = T - q * n;
#pragma unroll
    for (int j = 1; j <= kMaxPieces; ++j) {
        if (j >= n) {
            r.b[j] = T;
            continue;
        }
        const int x = j * q + (j < rem ? j : rem), lo = r.b[j - 1] + 4;   // ideal cut: piece lengths differ by at most one
        int best = r.b[j - 1], bd = 1 << 30;
        {   // inside the far block
            const int l = lo > 4 ? lo : 4;
            if (l <= fhi) {
                const int c = x < l ? l : (x > fhi ? fhi : x), d = c > x ? c - x : x - c;
                if (d < bd) { bd = d; best = c; }
            }
        }
        if (r.ntn > 0 && r.ntf >= lo) {   // the block boundary
            const int d = r.ntf > x ? r.ntf - x : x - r.ntf;
            if (d < bd) { bd = d; best = r.ntf; }
        }
        if (r.ntn > 0) {   // inside the near block
            int l = lo - r.ntf;
            l = l > 4 ? l : 4;
            if (l <= nhi) {
                const int xn = x - r.ntf, cn = xn < l ? l : (xn > nhi ? nhi : xn), c = r.ntf + cn, d = c > x ? c - x : x - c;
                if (d < bd) { bd = d; best = c; }
            }
        }
        r.b[j] = best;
    }
    return r;
}
// Range [t0, t1) of the far (which = 0) / near (1) block inside piece j; t1 <= t0: the piece has no part of that block.
__host__ __device__ inline void split_range(const SplitPair& r, int j, int which, int& t0, int& t1) {
    int lo = 0, hi = 0;
#pragma unroll
    for (int i = 0; i < kMaxPieces; ++i)
        if (i == j) {
            lo = r.b[i];
            hi = r.b[i + 1];
        }
    if (which == 0) {
        t0 = lo;
        t1 = hi < r.ntf ? hi : r.ntf;
    } else {
        t0 = (lo > r.ntf ? lo : r.ntf) - r.ntf;
        t1 = hi - r.ntf;
    }
}

// AULE_HIP_FWD_SPLIT=<n>: at most n pieces per pair; 0 (or 1) turns the path off (A/B measurements)
inline int split_max_pieces() {
    static const int v = [] {
        const char* e = getenv("AULE_HIP_FWD_SPLIT");
        if (e == nullptr || e[0] < '0' || e[0] > '9') return kMaxPieces;
        const int n = atoi(e);
        return n < kMaxPieces ? n : kMaxPieces;
    }();
    return v;
}

// AULE_HIP_FWD_SPLIT_MIN=<tiles>: the shortest piece (A/B; default kSplitMinTiles)
inline int split_min_tiles() {
    static const int v = [] {
        const char* e = getenv("AULE_HIP_FWD_SPLIT_MIN");
        const int n = (e != nullptr && e[0] >= '0' && e[0] <= '9') ? atoi(e) : kSplitMinTiles;
        return n < 4 ? 4 : n;
    }();
    return v;
}

struct SplitPlan {
    bool ok;
    int nqb, nwork, n;
    long long nitems;
    size_t bytes;
};
inline SplitPlan split_plan(const FwdArgs& a, int slots) {
    SplitPlan s{};
    const int pcoff = a.causal ? a.coff : kEverything;
    s.nqb = (a.Sq + kQBlock - 1) / kQBlock;
    s.nwork = a.causal ? (s.nqb + 1) / 2 : s.nqb;   // pairs of blocks, or single blocks
    const long long pairs = (long long)s.nwork * a.B * a.Hq;
    // as many pieces as still fit the chip in one round (one workgroup per CU), each at least kSplitMinTiles tiles of the longest
    // pair.  Measured in round 2 (profiles/r2_causal_split.txt): pieces of 17+ tiles win, 9-10 tiles are a wash at D = 128 and a loss at
    // D = 64: a piece costs a prologue, a partial row per query and its share of the merge launch.
    const int T = split_tiles(s.nqb - 1, a.Sk, pcoff) + (a.causal && s.nqb > 1 ? split_tiles(0, a.Sk, pcoff) : 0);
    long long n = slots / (pairs > 0 ? pairs : 1);
    n = n < split_max_pieces() ? n : split_max_pieces();
    n = n < T / split_min_tiles() ? n : T / split_min_tiles();
    // Two pieces that fill the chip need twice the length: going from half the CUs to all of them the busy ones lose ~a quarter of
    // their clock (the socket's power limit, profiles/r4_power_trace.txt), so 2 x 18 tiles on 256 workgroups is SLOWER than 36 on 128
    // once a piece's prologue, its partial rows and the merge launch are paid (round 4, one-wave-per-SIMD kernel, same box: B1 32q/8kv
    // S2048 63.6 us cut against 56.5 whole; non-causal B1 H8 S2048 47.4 against 42.0), while 2 x 34 (B1 H16 S4096: 86.9 against 90.1),
    // 2 x 66 (B1 H8 S8192: 135.8 against 148.4) and 4 x 17 on a quarter-full chip (B1 H8 S4096: 63.4 against 84.0) win.
    if (n == 2 && 2 * pairs > slots / 2 && T < 4 * kSplitMinTiles) n = 1;   // (the measured rule below keeps its constant: 64 tiles)
    s.n = (int)n;
    s.nitems = pairs * s.n;
    s.bytes = (size_t)s.n * a.B * a.Hq * a.Sq * (size_t)(a.D + kPartPad) * sizeof(float);
    if (s.n < 2) return s;
    int ncut = 0;   // pairs that do get cut
    for (int near = 0; near < s.nwork; ++near) {
        const SplitPair pr = split_cuts(a.causal ? s.nqb - 1 - near : near, near, a.Sk, pcoff, s.n, split_magic(s.n));
        int pieces = 0;
        for (int j = 0; j < s.n; ++j) pieces += pr.b[j + 1] > pr.b[j];
        ncut += pieces >= 2;
    }
    s.ok = 2 * ncut >= s.nwork;
    return s;
}

// What the merge kernel needs of the launch.
struct CombineParams {
    void* o;
    float* lse;
    const float* part;   // [npiece][part_rows][D + kPartPad]
    int part_rows;       // B * Hq * Sq
    int Sq, Sk, nqb, pair, pcoff, npiece;
    unsigned magic;
};

// Merge of the partial planes of every Q block the plan cut into ranges: one thread per four columns of a row, 1024 / D rows per
// 256-thread workgroup; blockIdx = (row group, Q block, b * Hq + h).  Blocks the plan left whole were finished by the forward
// kernel: their workgroups exit.  Bound: HBM (ranges x (D + 4) x 4 bytes read, D x 2 + 4 written per row).
template <class T, int D, int N>
__global__ void __launch_bounds__(256) fa_fwd_combine(const CombineParams p) {
    constexpr int TPR = D / 4, RPW = 256 / TPR, PP = D + kPartPad;
    const int bh = (int)blockIdx.z, qb = (int)blockIdx.y;
    const int mirror = p.pair ? p.nqb - 1 - qb : qb, near = qb < mirror ? qb : mirror, far = p.pair ? p.nqb - 1 - near : near;
    const int which = (qb == far) ? 0 : 1;
    const SplitPair pr = split_cuts(far, near, p.Sk, p.pcoff, p.npiece, p.magic);
    const int row = qb * kQBlock + (int)blockIdx.x * RPW + (int)threadIdx.x / TPR;
    if (row >= p.Sq) return;
    const int c4 = ((int)threadIdx.x % TPR) * 4;
    const size_t grow = (size_t)bh * p.Sq + row;
    unsigned mask = 0;   // which planes hold a range of this block (uniform over the workgroup: scalar code)
#pragma unroll
    for (int j = 0; j < kMaxPieces; ++j) {
        int t0, t1;
        split_range(pr, j, which, t0, t1);
        if (j < N && t1 > t0) mask |= 1u << j;
    }
    if (__builtin_popcount(mask) < 2) return;   // one range = the whole block: the forward kernel finished it
    // every plane's loads first (independent: one round trip), then the merge.  N = npiece is a template parameter and the loads are
    // unconditional -- a plane without a range of this block reads the first plane that has one, with weight 0 -- because
    // per-plane branches around the loads make the compiler copy the whole register array at every merge point.
    f32x4_t dj[N];
    f32x2_t ml[N];
    const float* const r0 = p.part + grow * PP;
    const size_t plane = (size_t)p.part_rows * PP;
    const int first = __builtin_ctz(mask);
#pragma unroll
    for (int j = 0; j < N; ++j) {
        const float* r = r0 + (size_t)((mask >> j) & 1u ? j : first) * plane;
        dj[j] = *reinterpret_cast<const f32x4_t*>(r + c4);
        ml[j] = *reinterpret_cast<const f32x2_t*>(r + D);
    }
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < N; ++j) mx = fmaxf(mx, ml[j][0]);   // (a stand-in plane repeats a real one: the maximum is unchanged)
    f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
    float lt = 0.f;
#pragma unroll
    for (int j = 0; j < N; ++j) {
        const float w = (mask >> j) & 1u ? fast_exp2(ml[j][0] - mx) : 0.f;
        acc += dj[j] * w;
        lt += ml[j][1] * w;
    }
    const float inv = 1.0f / lt;
    u32x2_t u;
    u[0] = T::pack2(acc[0] * inv, acc[1] * inv);
    u[1] = T::pack2(acc[2] * inv, acc[3] * inv);
    *reinterpret_cast<u32x2_t*>(static_cast<char*>(p.o) + (grow * D + c4) * 2) = u;
    if (c4 == 0 && p.lse != nullptr) p.lse[grow] = (mx + fast_log2(lt)) * kLn2;
}

template <class T, int D>
int launch_combine(const CombineParams& p, int B_Hq, hipStream_t stream) {
    constexpr int RPW = 256 / (D / 4);
    const dim3 cgrid(kQBlock / RPW, (unsigned)p.nqb, (unsigned)B_Hq);
    switch (p.npiece) {
        case 2: hipLaunchKernelGGL((fa_fwd_combine<T, D, 2>), cgrid, dim3(256), 0, stream, p); break;
        case 3: hipLaunchKernelGGL((fa_fwd_combine<T, D, 3>), cgrid, dim3(256), 0, stream, p); break;
        case 4: hipLaunchKernelGGL((fa_fwd_combine<T, D, 4>), cgrid, dim3(256), 0, stream, p); break;
        case 5: hipLaunchKernelGGL((fa_fwd_combine<T, D, 5>), cgrid, dim3(256), 0, stream, p); break;
        case 6: hipLaunchKernelGGL((fa_fwd_combine<T, D, 6>), cgrid, dim3(256), 0, stream, p); break;
        case 7: hipLaunchKernelGGL((fa_fwd_combine<T, D, 7>), cgrid, dim3(256), 0, stream, p); break;
        default: hipLaunchKernelGGL((fa_fwd_combine<T, D, 8>), cgrid, dim3(256), 0, stream, p); break;
    }
    return (int)hipGetLastError();
}

}  // namespace
}  // namespace aule_hip
