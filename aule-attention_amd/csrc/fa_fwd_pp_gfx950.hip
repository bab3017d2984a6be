// fa_fwd_pp_gfx950.hip -- ping-pong scheduled FlashAttention-2 forward (16-bit I/O).
//
// Same arithmetic, layouts and boundary as fa_fwd_gfx950.hip (read its header first);
// what changes is the SCHEDULE, designed around how a CDNA4 CU issues work:
//
//   * A 512-thread workgroup puts two wavefronts on each of the CU's 4 SIMDs (wave w and
//     wave w+4).  The matrix pipe and the VALU are separate pipes of a SIMD, but a
//     barrier-per-tile loop keeps both waves in the same phase (both in MFMAs, then both
//     in softmax VALU), so the pipes are used one after the other (measured: MFMA busy
//     30 %, SQ_WAIT_ANY 38 % of wave cycles -- profiles/r1_fwd_c2_v1kernel_rocprofv3_summary.txt).
//   * Here the per-tile work is split into two phases,
//         V-phase(j): softmax of S_j            (VALU only, no LDS, no MFMA)
//         M-phase(j): O += P_j V_j  and  S_{j+1} = K_{j+1} Q^T   (32 MFMAs + LDS reads)
//     and the two wave groups (waves 0-3 / waves 4-7) run ONE PHASE APART, re-aligned by
//     a workgroup barrier at every phase boundary: while one wave of a SIMD is in its
//     M-phase its partner is in its V-phase.  QK^T of the next tile is software-pipelined
//     into the PV phase of the current one so that each phase is either all-matrix or
//     all-vector.
//   * K is staged two tiles ahead and V one tile ahead (global -> VGPR at the start of the
//     V-phase, VGPR -> LDS at the end of the following M-phase), both double-buffered; the
//     hazard analysis is in DESIGN.md ("forward schedule").
//   * causal load balance: one workgroup processes the Q-block PAIR (i, n-1-i), so every
//     workgroup of a head does the same number of KV tiles.
#include <cstdlib>
#include <type_traits>

#include "fa_device.h"
#include "fa_kernels.h"
#include "fa_fwd_tile.h"

namespace aule_hip {
namespace {

struct FwdPPParams {
    const void* q;
    const void* k;
    const void* v;
    void* o;
    float* lse;
    int B, Hq, Hkv, Sq, Sk;
    float c;      // |scale| * log2(e)
    int negq;     // scale < 0
    int nqb;      // 256-row Q blocks
    int nwork;    // work items per head: ceil(nqb/2) when pairing, else nqb
    int pair;     // process Q blocks (i, nqb-1-i) in one workgroup
    int window;   // sliding window: key j visible to query i only if i - j < window (<= 0: off)
    int coff;     // causal position offset (query i sits at position i + coff; 0 = top-left rule)
    // SPLIT kernels only (short packed queries against long K/V, launch_pp_split): workgroup = (base work item, KV split)
    int nbase;          // base work items = nwork * B * Hq; blockIdx.x = split * nbase + base item
    int chunk;          // keys per split (multiple of kKVTile)
    int part_rows;      // rows of one partial plane = units * prow_per_unit
    int prow_per_unit;  // packed rows reserved per (batch, kv-head) unit (multiple of 32)
    int sq_orig;        // queries per head before packing (packed row r is query r % sq_orig of its head)
    float* part;        // [nsplit][part_rows][D + 2] fp32: un-normalised O, m (log2 units), l -- fa_fwd_splitkv_combine
    int dbg_flags;            // timeline build only: bit0 = group 1 computes nothing, bit1 = group 0 computes nothing
    unsigned long long* dbg;  // timeline build only: [8 waves][kTLMax] s_memtime stamps of workgroup 0
};

constexpr int kTLMax = 256;

// Which shapes take the SPLIT instances: to be set from measurements (tools/ppsplit_grid.py); until then every shape
// that passes the structural tests in pp_split_applicable() does.
#ifndef AULE_PPSPLIT_RULE
#define AULE_PPSPLIT_RULE true
#endif
#ifndef AULE_PPSPLIT_ROWS_PER_KEY
#define AULE_PPSPLIT_ROWS_PER_KEY 4   // total rows <= this x Sk (A/B builds override it: tools/ppsplit_edges.py)
#endif

// RAWOK: try the "fixed-reference" softmax first -- the row maximum of the FIRST tile stays the reference for
// the whole row, P = exp2(S*c - m_ref) (one v_fma + v_exp per element, the same arithmetic as the online
// form): no per-tile row maximum, no cross-half exchange, no O rescale test in the tile loop.  Exact algebra;
// valid while every row sum stays in [2^-100, 2^110], i.e. while no later logit exceeds the first tile's
// maximum by more than ~100 in log2 units (bf16 only: P needs the fp32 exponent range).  A Q block whose row
// sums leave that range is recomputed with the classic online softmax (workgroup-uniform decision).
// WIN: sliding window (the convention of triton_flash_amd.py:179-183: on top of the causal rule,
// key j is visible to query i only if i - j < window).  KV tiles entirely before the window of the Q block's first
// row are skipped by the whole workgroup; the online softmax tolerates rows whose keys in a tile are all masked.
template <class T, int D, bool CAUSAL, bool TL = false, bool RAWOK = false, bool WIN = false, bool SPLIT = false>
__global__ void __launch_bounds__(512) fa_fwd_pp_kernel(const FwdPPParams p) {
    static_assert(!(RAWOK && WIN), "the fixed-reference pass needs a visible key in the first tile of every row");
    static_assert(!SPLIT || !TL, "no timeline build of the SPLIT instances");
    static_assert(!(SPLIT && CAUSAL) || WIN, "a split can hide every key from a row: needs the -inf guards of the WIN softmax");
    using C = Cfg<D>;
    using v8 = typename T::v8;
    constexpr int RB = C::RB, RBP = C::RBP, CPR = C::CPR, KTILE = C::KTILE, VTILE = C::VTILE;
    constexpr int CH = C::CH, KS = C::KS, DB = C::DB;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const Ks = smem;
    char* const Vs = smem + 2 * KTILE;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2;  // 0: leads, 1: runs one phase behind
    const int l31 = lane & 31, hi = lane >> 5;
    char* const Qs = smem + 2 * KTILE + 2 * VTILE + wave * C::QSLAB;
    int* const redo_flag = reinterpret_cast<int*>(smem + C::LDS);
    if (RAWOK && tid < 2) redo_flag[tid] = 0;  // one verdict word per part (a workgroup has at most two)
    int tl_n = 0;
    auto stamp = [&]() __attribute__((always_inline)) {
        if constexpr (TL) {
            if (blockIdx.x == 0 && tl_n < kTLMax) {
                const unsigned long long t = __builtin_amdgcn_s_memtime();
                if (lane == 0) p.dbg[wave * kTLMax + tl_n] = t;
                ++tl_n;
            }
        }
    };

    // SPLIT: this workgroup sees keys [kv_begin, kv_begin + Sk) of its head as if they were the whole K/V (the
    // masks are position-independent), and leaves an un-normalised partial instead of O / LSE.
    const int split = SPLIT ? (int)(blockIdx.x / (unsigned)p.nbase) : 0;
    const WorkItem w = decode_work(SPLIT ? (int)(blockIdx.x % (unsigned)p.nbase) : (int)blockIdx.x, p.B, p.Hq, p.Hkv, p.nwork, false);
    const int kv_begin = SPLIT ? split * p.chunk : 0;
    const int Sq = p.Sq, Sk = SPLIT ? min(p.chunk, p.Sk - kv_begin) : p.Sk;
    const float c = p.c;

    const size_t kvhead = ((size_t)(w.b * p.Hkv + w.hk) * p.Sk + kv_begin) * RB;
    const __amdgpu_buffer_rsrc_t krs = make_srd(reinterpret_cast<const char*>(p.k) + kvhead, (unsigned)Sk * RB);
    const __amdgpu_buffer_rsrc_t vrs = make_srd(reinterpret_cast<const char*>(p.v) + kvhead, (unsigned)Sk * RB);

    // ---- staging maps (byte offsets inside one 64-row tile; the tile start goes in the SGPR offset).
    //      K: chunk c = tid + 512 i -> (row, 16-B chunk).
    //      V: 8 consecutive lanes fetch one [4 kv][16 d] sub-tile, so the LDS image is filled linearly
    //         by thread id (conflict-free ds_write_b128).
    int k_g[CH], k_lds[CH], v_g[CH];
#pragma unroll
    for (int i = 0; i < CH; ++i) {
        const int cidx = tid + 512 * i;
        const int row = cidx / CPR, cc = cidx % CPR;
        k_g[i] = row * RB + cc * 16;
        k_lds[i] = row * RBP + cc * 16;
        const int bidx = (tid >> 3) + 64 * i;  // sub-tile index = kv4 * (D/16) + d16
        v_g[i] = ((bidx / (D / 16)) * 4 + ((tid >> 1) & 3)) * RB + ((bidx % (D / 16)) * 2 + (tid & 1)) * 16;
    }
    const int ka_base = l31 * RBP + hi * 16;  // A operand (K) and B operand (Q): row l31, chunk 2ks + hi
    const int va_off = hi * (D / 16) * 128 + ((lane >> 4) & 1) * 128 + (lane & 15) * 8;

    u32x4_t kst[CH], vst[CH];
    int kvb = 0;  // first key of the first staged tile of the current Q block (window skipping)
    auto issue_k = [&](int kv0) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < CH; ++i)
            if (C::kFull || tid + 512 * i < C::NCHUNK)
                kst[i] = __builtin_amdgcn_raw_buffer_load_b128(krs, k_g[i], (kvb + kv0) * RB, 0);
    };
    auto issue_v = [&](int kv0) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < CH; ++i)
            if (C::kFull || tid + 512 * i < C::NCHUNK)
                vst[i] = __builtin_amdgcn_raw_buffer_load_b128(vrs, v_g[i], (kvb + kv0) * RB, 0);
    };
    auto write_k = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < CH; ++i)
            if (C::kFull || tid + 512 * i < C::NCHUNK)
                *reinterpret_cast<u32x4_t*>(Ks + buf * KTILE + k_lds[i]) = kst[i];
    };
    auto write_v = [&](int buf) __attribute__((always_inline)) {
        // The non-causal fixed-reference D = 128 kernels sit one register over the 256 of two waves per SIMD, and hipcc
        // spilled exactly this address (tid * 16): a scratch reload + s_waitcnt vmcnt(0) at the top of every tile
        // iteration.  There it is rebuilt from the lane id (two mbcnt) and the scalar wave id instead of staying live
        // across the loop: -0.3 .. -0.9 % in three same-box passes (tools/noncausal_ab.py).  Confined to exactly the
        // instantiations that spilled: applied to every non-causal kernel it made bf16 D = 64 3.2 % SLOWER (a kernel
        // that never spilled -- the extra instructions moved hipcc's allocation), and the causal kernels are left
        // byte-identical.
        int t16 = tid * 16;
        if constexpr (!CAUSAL && RAWOK && D == 128) {
            const int ln = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
            t16 = (ln << 4) + wave * 1024;
        }
#pragma unroll
        for (int i = 0; i < CH; ++i)
            if (C::kFull || tid + 512 * i < C::NCHUNK)
                *reinterpret_cast<u32x4_t*>(Vs + buf * VTILE + t16 + i * 8192) = vst[i];
    };

    const int nparts = (p.pair && (p.nqb - 1 - w.blk) != w.blk) ? 2 : 1;
    for (int part = 0; part < nparts; ++part) {
        const int qb = p.pair ? (part == 0 ? p.nqb - 1 - w.blk : w.blk) : w.blk;
        const int q0w = qb * kQBlock + wave * 32;
        const int qrow = q0w + l31;

        const int coff = p.coff;                  // position of query row i = i + coff
        // SPLIT: rows are packed (row r = query r % sq_orig of some head of the group) and key indices are local to
        // this split, so positions are taken per lane and the block / wave bounds cover every head's queries
        const int qposv = SPLIT ? (qrow % p.sq_orig) + coff - kv_begin : qrow + coff;      // this lane's query position
        const int blk_pos_lo = SPLIT ? coff - kv_begin : qb * kQBlock + coff;               // first position in the block
        const int kv_hi = CAUSAL ? max(1, min(Sk, SPLIT ? p.sq_orig + coff - kv_begin : qb * kQBlock + kQBlock + coff)) : Sk;
        int t_lo = 0;  // first tile any row of this Q block can see
        if constexpr (WIN) t_lo = min(max(0, blk_pos_lo - p.window + 1) / kKVTile, (kv_hi + kKVTile - 1) / kKVTile - 1);
        kvb = t_lo * kKVTile;
        const int nt = (kv_hi + kKVTile - 1) / kKVTile - t_lo;   // tiles staged by the workgroup (>= 1)
        const int wave_kv_hi = CAUSAL ? (SPLIT ? kv_hi : min(Sk, q0w + 32 + coff)) : Sk;  // keys visible to this wave
        int na = max(1, (wave_kv_hi + kKVTile - 1) / kKVTile - t_lo);  // tiles this wave computes (a prefix)
        if constexpr (TL) {
            if ((p.dbg_flags >> grp) & 1) na = 0;
        }
        f32x16_t o[DB];
        float m, l;

        auto run_part = [&](auto raw_tag) __attribute__((always_inline)) {
            constexpr bool RAW = decltype(raw_tag)::value != 0;
            // Every tile the prologue needs is requested up front (one HBM round trip, not two): K_0, V_0, K_1 by
            // all waves, V_1 and K_2 by group 1 (see the entry state below); loads past the end of K/V read 0.
            issue_k(0);
            issue_v(0);
            u32x4_t kpre1[CH], vpre1[CH], kpre2[CH];
#pragma unroll
            for (int i = 0; i < CH; ++i)
                if (C::kFull || tid + 512 * i < C::NCHUNK) {
                    kpre1[i] = __builtin_amdgcn_raw_buffer_load_b128(krs, k_g[i], (kvb + kKVTile) * RB, 0);
                    if (grp == 1) {
                        vpre1[i] = __builtin_amdgcn_raw_buffer_load_b128(vrs, v_g[i], (kvb + kKVTile) * RB, 0);
                        kpre2[i] = __builtin_amdgcn_raw_buffer_load_b128(krs, k_g[i], (kvb + 2 * kKVTile) * RB, 0);
                    }
                }
            // ---- Q fragments (B operand of S^T = K.Q^T) in registers: lane (q, hi) holds d = 16ks+8hi..+7.
            //      (An earlier version re-read Q from LDS per tile because the kernel did not fit 256 VGPRs; with
            //      the straight-line loop it uses ~190, so the 32 registers are affordable and save 8 of the 24
            //      ds_read_b128 of every QK^T phase.)  Rows >= Sq read as 0 (buffer bounds check).
            v8 qf[KS];
            {
                const size_t qhead = (size_t)(w.b * p.Hq + w.h) * Sq * RB;
                const __amdgpu_buffer_rsrc_t qrs = make_srd(reinterpret_cast<const char*>(p.q) + qhead, (unsigned)Sq * RB);
                const unsigned flip = p.negq ? 0x80008000u : 0u;
                u32x4_t qx[KS];
#pragma unroll
                for (int ks = 0; ks < KS; ++ks)
                    qx[ks] = __builtin_amdgcn_raw_buffer_load_b128(qrs, qrow * RB + (2 * ks + hi) * 16, 0, 0);
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    u32x4_t x = qx[ks];
                    x[0] ^= flip; x[1] ^= flip; x[2] ^= flip; x[3] ^= flip;
                    qf[ks] = as_v8<T>(x);
                }
            }

#pragma unroll
            for (int d = 0; d < DB; ++d)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
            m = -INFINITY;
            l = 0.f;
            f32x16_t s[2];
            v8 pb[2][2];

            // Both MFMA loops are software-pipelined by hand: a wave issues in order, so an MFMA whose LDS
            // operands were requested just before it stalls for the whole LDS latency (measured: 16 MFMAs
            // took ~950 cycles instead of 512).  Operands are requested kAhead steps early (about two MFMA
            // operands ahead is the measured optimum for both loops: deeper look-ahead costs 3-4 %); the
            // sched_group_barrier sequence pins "1 MFMA, then the reads of a later step" in the final code.
            auto qk = [&](int buf) __attribute__((always_inline)) {  // S^T = K_tile . Q^T   (all LDS offsets are immediates)
                const char* kb = Ks + buf * KTILE + ka_base;
                constexpr int kAhead = 1;
                u32x4_t kf[KS][2];
                auto rd = [&](int ks) __attribute__((always_inline)) {
                    kf[ks][0] = *reinterpret_cast<const u32x4_t*>(kb + ks * 32);
                    kf[ks][1] = *reinterpret_cast<const u32x4_t*>(kb + ks * 32 + 32 * RBP);
                };
                f32x16_t z;
#pragma unroll
                for (int r = 0; r < 16; ++r) z[r] = 0.f;
#pragma unroll
                for (int ks = 0; ks < kAhead && ks < KS; ++ks) rd(ks);
                __builtin_amdgcn_sched_group_barrier(0x100, 2 * (kAhead < KS ? kAhead : KS), 0);
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    if (ks + kAhead < KS) rd(ks + kAhead);
                    s[0] = T::mfma(as_v8<T>(kf[ks][0]), qf[ks], ks == 0 ? z : s[0]);
                    s[1] = T::mfma(as_v8<T>(kf[ks][1]), qf[ks], ks == 0 ? z : s[1]);
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    if (ks + kAhead < KS) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                }
            };
            auto pv = [&](int buf) __attribute__((always_inline)) {  // O^T += V^T . P^T
                const char* vb = Vs + buf * VTILE + va_off;
                constexpr int NST = 4 * DB;  // MFMA steps: (sb, kk) outer, d inner
                constexpr int kAhead = 2;
                s16x4_t a0[NST], a1[NST];
                auto rd = [&](int st) __attribute__((always_inline)) {
                    const int sk = st / DB, d = st % DB;  // sk = 2*sb + kk
                    const int off = ((4 * sk) * (D / 16) + 2 * d) * 128;
                    a0[st] = lds_tr16(vb + off);
                    a1[st] = lds_tr16(vb + off + 2 * (D / 16) * 128);
                };
#pragma unroll
                for (int st = 0; st < kAhead && st < NST; ++st) rd(st);
                __builtin_amdgcn_sched_group_barrier(0x100, 2 * (kAhead < NST ? kAhead : NST), 0);
#pragma unroll
                for (int st = 0; st < NST; ++st) {
                    if (st + kAhead < NST) rd(st + kAhead);
                    const int sk = st / DB, d = st % DB;
                    o[d] = T::mfma(as_v8<T>(a0[st], a1[st]), pb[sk >> 1][sk & 1], o[d]);
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    if (st + kAhead < NST) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                }
            };
            auto softmax = [&](int kv0, auto fixed_tag) __attribute__((always_inline)) {
                constexpr bool FIXED = decltype(fixed_tag)::value != 0;  // S_j -> P_j (16-bit, in registers); updates m, l, o
                const bool need_mask = (CAUSAL && (kv0 + kKVTile - 1 > (SPLIT ? blk_pos_lo : q0w + coff))) || (kv0 + kKVTile > Sk) ||
                                       (WIN && ((SPLIT ? blk_pos_lo + p.sq_orig - 1 : q0w + coff + 31) - kv0 >= p.window));
                if (need_mask) {
#pragma unroll
                    for (int sb = 0; sb < 2; ++sb)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int kv = kv0 + sb * 32 + crow(r, hi);
                            const bool vis = (kv < Sk) && (!CAUSAL || kv <= qposv) && (!WIN || qposv - kv < p.window);
                            s[sb][r] = vis ? s[sb][r] : -INFINITY;
                        }
                }
                if constexpr (FIXED) {
                    // pinned single-issue forms: as plain IR hipcc packs the pairs into v_pk_fma_f32 /
                    // v_pk_add_f32, which cost more than two scalar instructions next to the partner's MFMAs
                    const float nm = -m;
                    float a0 = 0.f, a1 = 0.f;
                    u32x4_t pr[2][2];
#pragma unroll
                    for (int sb = 0; sb < 2; ++sb)
#pragma unroll
                        for (int kk = 0; kk < 2; ++kk)
                            pr[sb][kk] = softmax_oct<T>(s[sb][8 * kk], s[sb][8 * kk + 1], s[sb][8 * kk + 2], s[sb][8 * kk + 3],
                                                          s[sb][8 * kk + 4], s[sb][8 * kk + 5], s[sb][8 * kk + 6], s[sb][8 * kk + 7],
                                                          c, nm, a0, a1);
                    l += a0 + a1;  // (two asm adds behind the last v_exp)
                    asm volatile("" : "+v"(pr[0][0]), "+v"(pr[0][1]), "+v"(pr[1][0]), "+v"(pr[1][1]), "+v"(l));
#pragma unroll
                    for (int sb = 0; sb < 2; ++sb)
#pragma unroll
                        for (int kk = 0; kk < 2; ++kk) pb[sb][kk] = as_v8<T>(pr[sb][kk]);
                    return;
                }
                // row max: four independent v_max3_f32 chains (fmaxf() costs an extra canonicalising v_max per
                // MFMA output, and one 31-deep chain is latency-bound)
                float mx4[4];
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    const int sb = q4 >> 1, b0 = 8 * (q4 & 1);
                    mx4[q4] = max3(s[sb][b0], s[sb][b0 + 1], s[sb][b0 + 2]);
                    mx4[q4] = max3(mx4[q4], s[sb][b0 + 3], s[sb][b0 + 4]);
                    mx4[q4] = max3(mx4[q4], s[sb][b0 + 5], s[sb][b0 + 6]);
                }
                float mx = max3(mx4[0], mx4[1], s[0][7]);
                mx = max3(mx, mx4[2], s[0][15]);
                mx = max3(mx, mx4[3], s[1][7]);
                mx = fmaxf(mx, s[1][15]);
                mx = fmaxf(mx, xhalf_fast(mx));
                const float mxc = mx * c;
                stamp();
                // lazy rescale (exact algebra, different rounding): the running max is only raised -- and O, l
                // rescaled -- when some row's new maximum exceeds the kept one by more than 2^kRescaleThr;
                // otherwise P is formed against the kept max (P <= 2^8, fine for bf16/fp16 and fp32 sums).
                if (__builtin_amdgcn_ballot_w64(mxc > m + kRescaleThr) != 0) {
                    const float m_new = fmaxf(m, mxc);
                    float alpha = fast_exp2(m - m_new);
                    if constexpr (WIN) alpha = (m_new == -INFINITY) ? 1.0f : alpha;  // row still without a visible key
                    m = m_new;
                    l *= alpha;
#pragma unroll
                    for (int d = 0; d < DB; ++d)
#pragma unroll
                        for (int r = 0; r < 16; ++r) o[d][r] *= alpha;
                }
                // the same fused statement as the fixed-reference form, against the (lazily raised) running maximum
                const float m_sub = (WIN && m == -INFINITY) ? 0.f : m;  // (-inf) - (-inf) would be NaN; P = exp2(-inf) = 0
                const float nmv = -m_sub;
                float a0 = 0.f, a1 = 0.f;
                u32x4_t pu[2][2];
#pragma unroll
                for (int sb = 0; sb < 2; ++sb)
#pragma unroll
                    for (int kk = 0; kk < 2; ++kk)
                        pu[sb][kk] = softmax_oct<T>(s[sb][8 * kk], s[sb][8 * kk + 1], s[sb][8 * kk + 2], s[sb][8 * kk + 3],
                                                    s[sb][8 * kk + 4], s[sb][8 * kk + 5], s[sb][8 * kk + 6], s[sb][8 * kk + 7],
                                                    c, nmv, a0, a1);
                const f32x2_t lt2 = {a0, a1};
                l += lt2[0] + lt2[1];
                // Pin the results of this phase HERE: the softmax is register-only code that LLVM otherwise
                // sinks past the barrier into the block that consumes P (next to this wave's own MFMAs).
                asm volatile("" : "+v"(pu[0][0]), "+v"(pu[0][1]), "+v"(pu[1][0]), "+v"(pu[1][1]), "+v"(l), "+v"(m));
#pragma unroll
                for (int sb = 0; sb < 2; ++sb)
#pragma unroll
                    for (int kk = 0; kk < 2; ++kk) pb[sb][kk] = as_v8<T>(pu[sb][kk]);
            };

            // ---- prologue.  Staging rule of the main loop: ALL global->LDS staging happens in the V-phases
            //      (the VALU-bound phase, whose LDS/VMEM issue ports are idle), never in the M-phases:
            //      in V-phase(t) group d (0 or 1) first writes the tiles it loaded one phase earlier,
            //      V_{t+d} and K_{t+1+d}, then requests V_{t+1+d} and K_{t+2+d}.  Entry state for t = 0:
            //      K_0 in LDS; group 0 holds (V_0, K_1) in registers, group 1 has written its share of
            //      (V_0, K_1) and holds (V_1, K_2).  Hazard analysis: DESIGN.md "forward schedule".
            stamp();  // TL: part start (loads issued)
            write_k(0);
            stamp();  // TL: K_0/V_0/Q arrived
#pragma unroll
            for (int i = 0; i < CH; ++i) kst[i] = kpre1[i];  // K_1
            if (grp == 1) {
                write_v(0);
                if (nt > 1) write_k(1);
#pragma unroll
                for (int i = 0; i < CH; ++i) {
                    vst[i] = vpre1[i];  // V_1
                    kst[i] = kpre2[i];  // K_2
                }
            }
            __syncthreads();
            stamp();  // TL: prologue barrier passed
            if (grp == 1) __syncthreads();  // group 1 starts one phase late
            if (na > 0) qk(0);              // pre-phase: S_0
            stamp();  // TL: pre-phase done
            __builtin_amdgcn_sched_barrier(0);
            __syncthreads();
            __builtin_amdgcn_sched_barrier(0);

            // One tile step = V-phase + barrier + M-phase + barrier.  MODE is a compile-time constant so that
            // the steady-state loop body is straight-line code (a per-iteration branch on `na` made hipcc
            // copy the 64 O accumulators at every merge point): 2 = softmax, PV and next QK^T; 1 = softmax
            // and PV (this wave's last active tile); 0 = fully masked tile, only staging and barriers.
            // fixed-reference pass: every wave posts its range verdict (NaN fails it too) BEFORE the barrier that
            // re-aligns the two groups, so the workgroup-uniform decision costs no barrier of its own
            auto flag_range = [&]() __attribute__((always_inline)) {
                const float lsum = l + xhalf(l);
                const bool ok = (na == 0) || ((lsum > 0x1p-100f) && (lsum < 0x1p110f));
                if (__builtin_amdgcn_ballot_w64(!ok) != 0 && lane == 0) redo_flag[part] = 1;
            };
            auto tile_step = [&](int j, auto mode_tag, auto fixed_tag) __attribute__((always_inline)) {
                constexpr int MODE = decltype(mode_tag)::value;
                // ---- V-phase(j): stage (see the prologue comment), then softmax(S_j)
                stamp();
                if (j + grp < nt) write_v((j + grp) & 1);
                if (j + 1 + grp < nt) write_k((j + 1 + grp) & 1);
                if (j + 1 + grp < nt) issue_v((j + 1 + grp) * kKVTile);
                if (j + 2 + grp < nt) issue_k((j + 2 + grp) * kKVTile);
                __builtin_amdgcn_s_setprio(AULE_VPRIO);
                if constexpr (MODE >= 1) softmax(kvb + j * kKVTile, fixed_tag);
                __builtin_amdgcn_s_setprio(0);
                stamp();
                // phase boundary: nothing may move across (hipcc would interleave this wave's softmax with
                // its own MFMAs -- measured with tools/timeline.py -- which defeats the group alternation)
                __builtin_amdgcn_sched_barrier(0);
                __syncthreads();
                __builtin_amdgcn_sched_barrier(0);
                stamp();
                // ---- M-phase(j): O += P_j V_j ; S_{j+1} = K_{j+1} Q^T   (MFMA + LDS reads only)
                __builtin_amdgcn_s_setprio(AULE_MPRIO);
                if constexpr (MODE >= 1) pv(j & 1);
                if constexpr (MODE == 2) {
                    __builtin_amdgcn_sched_barrier(0);  // P dies after PV, S is born in QK: do not overlap them
                    if constexpr (TL) { keep_live(o[0], o[DB - 1]); stamp(); }
                    qk((j + 1) & 1);
                    if constexpr (TL) { keep_live(s[0], s[1]); stamp(); }
                }
                __builtin_amdgcn_s_setprio(0);
                stamp();
                if constexpr (RAW && MODE <= 1) {
                    if (grp == 1 && j == nt - 1) flag_range();  // group 1: last barrier of the part follows
                }
                __builtin_amdgcn_sched_barrier(0);
                __syncthreads();
                __builtin_amdgcn_sched_barrier(0);
            };
            using std::integral_constant;
            int j = 0;
            if constexpr (RAW) {  // tile 0 sets the reference with the online form; every later tile keeps it
                if (na > 1) { tile_step(0, integral_constant<int, 2>{}, integral_constant<int, 0>{}); j = 1; }
            }
            for (; j + 1 < na; ++j) tile_step(j, integral_constant<int, 2>{}, integral_constant<int, RAW ? 1 : 0>{});
            if (j < na) {
                if (RAW && j > 0) tile_step(j, integral_constant<int, 1>{}, integral_constant<int, 1>{});
                else tile_step(j, integral_constant<int, 1>{}, integral_constant<int, 0>{});
                ++j;
            }
            for (; j < nt; ++j) tile_step(j, integral_constant<int, 0>{}, integral_constant<int, 0>{});

            stamp();  // TL: end of tile loop
            if constexpr (RAW) {
                if (grp == 0) flag_range();
            }
            if (grp == 0) __syncthreads();  // pairs with group 1's last phase barrier: all waves aligned again
            stamp();  // TL: aligned
        };
        if constexpr (RAWOK) {
            run_part(std::integral_constant<int, 1>{});
            const int redo = redo_flag[part];
            stamp();  // TL: range check done
            if (redo) {
                run_part(std::integral_constant<int, 0>{});
            }
        } else {
            run_part(std::integral_constant<int, 0>{});
        }

        // ---- epilogue: O = O^T / l, transposed through this wave's (now idle) Q slab so that the
        //      global stores are whole 16-byte chunks of full rows (the direct form is 16 row-strided
        //      8-byte stores per lane and was ~16k cycles per Q block); LSE = (m + log2 l) * ln2
        const float lt = l + xhalf(l);
        if constexpr (SPLIT) {
            // partial (O^T un-normalised, m, l) in fp32; rows go through the wave's Q slab like the O epilogue below,
            // half a row (D/2 floats = RB bytes) per pass, so that the global stores are whole rows
            float* const prow0 = p.part + ((size_t)split * p.part_rows + (size_t)(w.b * p.Hkv + w.hk) * p.prow_per_unit) * (D + 2);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
#pragma unroll
                for (int d = 0; d < DB; ++d)
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        constexpr int HALF = D / 2;
                        const int d0 = 32 * d + 8 * g4;   // this lane holds columns d0 + 4 hi .. + 3
                        if (d0 / HALF == h) {
                            const f32x4_t x = {o[d][4 * g4 + 0], o[d][4 * g4 + 1], o[d][4 * g4 + 2], o[d][4 * g4 + 3]};
                            *reinterpret_cast<f32x4_t*>(Qs + l31 * RBP + ((d0 % HALF) + 4 * hi) * 4) = x;
                        }
                    }
#pragma unroll
                for (int i = 0; i < (32 * CPR) / 64; ++i) {
                    const int cidx = lane + 64 * i;
                    const int row = cidx / CPR, cc = cidx % CPR;
                    const f32x4_t x = *reinterpret_cast<const f32x4_t*>(Qs + row * RBP + cc * 16);
                    if (q0w + row < Sq) {
                        float* dst = prow0 + (size_t)(q0w + row) * (D + 2) + h * (D / 2) + cc * 4;
                        // rows are (D + 2) floats apart: 8-byte aligned, not 16
                        *reinterpret_cast<f32x2_t*>(dst) = f32x2_t{x[0], x[1]};
                        *reinterpret_cast<f32x2_t*>(dst + 2) = f32x2_t{x[2], x[3]};
                    }
                }
            }
            if (qrow < Sq && hi == 0) {
                prow0[(size_t)qrow * (D + 2) + D] = m;
                prow0[(size_t)qrow * (D + 2) + D + 1] = lt;
            }
            continue;
        }
        const float inv = (WIN && !(lt > 0.f)) ? 0.f : 1.0f / lt;  // a row with no visible key at all: O = 0, LSE = -inf
#pragma unroll
        for (int d = 0; d < DB; ++d)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                u32x2_t u;
                u[0] = T::pack2(o[d][4 * g4 + 0] * inv, o[d][4 * g4 + 1] * inv);
                u[1] = T::pack2(o[d][4 * g4 + 2] * inv, o[d][4 * g4 + 3] * inv);
                *reinterpret_cast<u32x2_t*>(Qs + l31 * RBP + (32 * d + 8 * g4 + 4 * hi) * 2) = u;
            }
        // (wave-private LDS: program order + the compiler's lgkmcnt wait are enough, no barrier)
        {
            char* obase = reinterpret_cast<char*>(p.o) + ((size_t)(w.b * p.Hq + w.h) * Sq) * RB;
#pragma unroll
            for (int i = 0; i < (32 * CPR) / 64; ++i) {
                const int cidx = lane + 64 * i;
                const int row = cidx / CPR, cc = cidx % CPR;
                const u32x4_t x = *reinterpret_cast<const u32x4_t*>(Qs + row * RBP + cc * 16);
                if (q0w + row < Sq) *reinterpret_cast<u32x4_t*>(obase + (size_t)(q0w + row) * RB + cc * 16) = x;
            }
        }
        if (qrow < Sq && p.lse != nullptr && hi == 0)
            p.lse[(size_t)(w.b * p.Hq + w.h) * Sq + qrow] = (m + fast_log2(lt)) * kLn2;
        stamp();  // TL: epilogue issued
    }
}

// AULE_HIP_FWD_SOFTMAX = "raw" (default for bf16) | "classic" (always the online softmax; A/B measurements)
static bool raw_softmax_enabled() {
    static const int v = [] {
        const char* e = getenv("AULE_HIP_FWD_SOFTMAX");
        return (e != nullptr && e[0] == 'c') ? 0 : 1;
    }();
    return v == 1;
}

template <class T, int D>
int launch_pp(const FwdArgs& a, hipStream_t stream) {
    FwdPPParams p;
    p.q = a.q; p.k = a.k; p.v = a.v; p.o = a.o; p.lse = a.lse;
    p.B = a.B; p.Hq = a.Hq; p.Hkv = a.Hkv; p.Sq = a.Sq; p.Sk = a.Sk;
    float c = a.scale * kLog2e;
    p.negq = c < 0.f;
    c = c < 0.f ? -c : c;
    if (c == 0.f) c = 1e-30f;
    p.c = c;
    p.nqb = (a.Sq + kQBlock - 1) / kQBlock;
    p.pair = a.causal ? 1 : 0;
    p.nwork = p.pair ? (p.nqb + 1) / 2 : p.nqb;
    p.dbg = nullptr;
    p.dbg_flags = 0;
    p.window = a.window > 0 ? a.window : 0;
    p.coff = a.causal ? a.coff : 0;
    p.sq_orig = a.Sq; p.nbase = 1; p.chunk = 0; p.part_rows = 0; p.prow_per_unit = 0; p.part = nullptr;
    const dim3 grid((unsigned)(p.nwork * a.B * a.Hq)), block(512);
    const size_t lds = Cfg<D>::LDS + 16;
    if (p.window > 0) {  // sliding window: online softmax only
        if (a.causal)
            hipLaunchKernelGGL((fa_fwd_pp_kernel<T, D, true, false, false, true>), grid, block, lds, stream, p);
        else
            hipLaunchKernelGGL((fa_fwd_pp_kernel<T, D, false, false, false, true>), grid, block, lds, stream, p);
        return (int)hipGetLastError();
    }
    if constexpr (std::is_same<T, Bf16Traits>::value) {
        if (raw_softmax_enabled()) {
            if (a.causal)
                hipLaunchKernelGGL((fa_fwd_pp_kernel<T, D, true, false, true>), grid, block, lds, stream, p);
            else
                hipLaunchKernelGGL((fa_fwd_pp_kernel<T, D, false, false, true>), grid, block, lds, stream, p);
            return (int)hipGetLastError();
        }
    }
    if (a.causal)
        hipLaunchKernelGGL((fa_fwd_pp_kernel<T, D, true>), grid, block, lds, stream, p);
    else
        hipLaunchKernelGGL((fa_fwd_pp_kernel<T, D, false>), grid, block, lds, stream, p);
    return (int)hipGetLastError();
}

// ---- short packed queries against long K/V on the tiled kernel (SPLIT instances) -------------------------------
// Non-causal, no window: masks do not depend on the query position, so (1) the g = Hq/Hkv query heads that share a
// K/V head are contiguous in Q / O / LSE and can be read as ONE head with g*Sq rows (B' = B*Hkv, Hq' = Hkv' = 1):
// a 256-row Q block is then full of real rows and K/V is streamed once per group; (2) the key range is cut into
// `nsplit` chunks, one workgroup each, so that B'*nqb'*nsplit workgroups fill the chip; every workgroup leaves an
// fp32 partial and fa_fwd_splitkv_combine (fa_fwd_splitkv_gfx950.hip) merges them.  Complements the wave-per-chunk
// split-KV kernel, which wins while a unit has few packed rows (it is HBM-bound; this one is MFMA-bound).
struct PPSplitPlan {
    int g, rows, nqb, nbase, ntiles, nsplit, chunk, nrt;
};

static PPSplitPlan pp_split_plan(const FwdArgs& a) {
    PPSplitPlan s;
    s.g = a.Hq / a.Hkv;
    s.rows = s.g * a.Sq;
    s.nqb = (s.rows + kQBlock - 1) / kQBlock;
    s.nbase = a.B * a.Hkv * s.nqb;
    s.ntiles = (a.Sk + kKVTile - 1) / kKVTile;
    int want = 256 / (s.nbase > 0 ? s.nbase : 1);          // one workgroup per CU (137 KB of LDS each)
    const int most = s.ntiles / 4;                          // at least 4 tiles per split: the prologue costs ~3
    if (want > most) want = most;
    if (want < 1) want = 1;
    const int tiles_per = (s.ntiles + want - 1) / want;
    s.chunk = tiles_per * kKVTile;
    s.nsplit = (a.Sk + s.chunk - 1) / s.chunk;
    s.nrt = (s.rows + 31) / 32;
    return s;
}

template <class T, int D>
int launch_pp_split(const FwdArgs& a, hipStream_t stream) {
    const PPSplitPlan s = pp_split_plan(a);
    FwdPPParams p;
    p.q = a.q; p.k = a.k; p.v = a.v; p.o = nullptr; p.lse = nullptr;
    p.B = a.B * a.Hkv; p.Hq = 1; p.Hkv = 1; p.Sq = s.rows; p.Sk = a.Sk;
    float c = a.scale * kLog2e;
    p.negq = c < 0.f;
    c = c < 0.f ? -c : c;
    if (c == 0.f) c = 1e-30f;
    p.c = c;
    p.nqb = s.nqb; p.pair = 0; p.nwork = s.nqb;
    p.dbg = nullptr; p.dbg_flags = 0; p.window = 0; p.coff = 0;
    p.sq_orig = a.Sq;
    p.nbase = s.nbase; p.chunk = s.chunk;
    p.prow_per_unit = s.nrt * 32;
    p.part_rows = a.B * a.Hkv * p.prow_per_unit;
    const size_t bytes = (size_t)s.nsplit * p.part_rows * (D + 2) * sizeof(float);
    if (a.query_ws != nullptr) {
        *a.query_ws = bytes;
        return 0;
    }
    ScopedWorkspace ws(bytes, a.ws, a.ws_bytes, stream);   // caller's buffer, or stream-ordered like the split-KV kernel's
    if (ws.err != hipSuccess) return (int)ws.err;
    p.part = static_cast<float*>(ws.ptr);
    const dim3 grid((unsigned)(s.nbase * s.nsplit)), block(512);
    const size_t lds = Cfg<D>::LDS + 16;
    if (a.causal) {   // bottom-right chunk: guarded online softmax (rows can be fully masked inside a split), no window
        p.coff = a.coff;
        p.window = 0x3fffffff;
        hipLaunchKernelGGL((fa_fwd_pp_kernel<T, D, true, false, false, true, true>), grid, block, lds, stream, p);
    } else {
        bool raw = false;
        if constexpr (std::is_same<T, Bf16Traits>::value) {
            raw = raw_softmax_enabled();
            if (raw) hipLaunchKernelGGL((fa_fwd_pp_kernel<T, D, false, false, true, false, true>), grid, block, lds, stream, p);
        }
        if (!raw) hipLaunchKernelGGL((fa_fwd_pp_kernel<T, D, false, false, false, false, true>), grid, block, lds, stream, p);
    }
    int rc = (int)hipGetLastError();
    if (rc == 0) rc = launch_splitkv_combine(a, p.part, s.nsplit, s.nrt, stream);
    return rc;
}

template <class T, int D>
int set_attr_pp() {
    int rc = (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&fa_fwd_pp_kernel<T, D, true>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, Cfg<D>::LDS + 16);
    rc |= (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&fa_fwd_pp_kernel<T, D, false>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, Cfg<D>::LDS + 16);
    rc |= (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&fa_fwd_pp_kernel<T, D, true, false, false, true>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, Cfg<D>::LDS + 16);
    rc |= (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&fa_fwd_pp_kernel<T, D, false, false, false, true>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, Cfg<D>::LDS + 16);
    rc |= (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&fa_fwd_pp_kernel<T, D, false, false, false, false, true>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, Cfg<D>::LDS + 16);
    rc |= (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&fa_fwd_pp_kernel<T, D, true, false, false, true, true>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, Cfg<D>::LDS + 16);
    if constexpr (std::is_same<T, Bf16Traits>::value) {
        rc |= (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&fa_fwd_pp_kernel<T, D, false, false, true, false, true>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, Cfg<D>::LDS + 16);
        rc |= (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&fa_fwd_pp_kernel<T, D, true, false, true>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, Cfg<D>::LDS + 16);
        rc |= (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&fa_fwd_pp_kernel<T, D, false, false, true>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, Cfg<D>::LDS + 16);
    }
    return rc;
}

}  // namespace

#ifdef AULE_DEBUG_HOOKS
// Debug: run the bf16 D=128 kernel with s_memtime stamps (4 per tile per wave, workgroup 0):
// [V-phase start, V-phase end, M-phase start (after barrier), M-phase end].
int launch_fwd_pp_timeline(const FwdArgs& a, unsigned long long* dbg, hipStream_t stream) {
    if (a.dtype != kBF16 || a.D != 128) return -1;
    FwdPPParams p;
    p.q = a.q; p.k = a.k; p.v = a.v; p.o = a.o; p.lse = a.lse;
    p.B = a.B; p.Hq = a.Hq; p.Hkv = a.Hkv; p.Sq = a.Sq; p.Sk = a.Sk;
    p.c = a.scale * kLog2e; p.negq = 0;
    p.nqb = (a.Sq + kQBlock - 1) / kQBlock;
    p.pair = a.causal ? 1 : 0;
    p.nwork = p.pair ? (p.nqb + 1) / 2 : p.nqb;
    p.window = 0;
    p.coff = 0;
    p.sq_orig = a.Sq; p.nbase = 1; p.chunk = 0; p.part_rows = 0; p.prow_per_unit = 0; p.part = nullptr;
    p.dbg = dbg;
    p.dbg_flags = getenv("AULE_TL_FLAGS") ? atoi(getenv("AULE_TL_FLAGS")) : 0;
    const dim3 grid((unsigned)(p.nwork * a.B * a.Hq)), block(512);
    const size_t lds = Cfg<128>::LDS + 16;
    auto go = [&](auto kern) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        hipLaunchKernelGGL(kern, grid, block, lds, stream, p);
    };
    if (raw_softmax_enabled()) {
        if (a.causal) go(&fa_fwd_pp_kernel<Bf16Traits, 128, true, true, true>);
        else go(&fa_fwd_pp_kernel<Bf16Traits, 128, false, true, true>);
    } else {
        if (a.causal) go(&fa_fwd_pp_kernel<Bf16Traits, 128, true, true, false>);
        else go(&fa_fwd_pp_kernel<Bf16Traits, 128, false, true, false>);
    }
    return (int)hipGetLastError();
}

#endif  // AULE_DEBUG_HOOKS

int launch_fwd_pp(const FwdArgs& a, hipStream_t stream) {
    if (a.dtype == kBF16) {
        if (a.D == 128) return launch_pp<Bf16Traits, 128>(a, stream);
        if (a.D == 64) return launch_pp<Bf16Traits, 64>(a, stream);
        if (a.D == 32) return launch_pp<Bf16Traits, 32>(a, stream);
    } else if (a.dtype == kF16) {
        if (a.D == 128) return launch_pp<F16Traits, 128>(a, stream);
        if (a.D == 64) return launch_pp<F16Traits, 64>(a, stream);
        if (a.D == 32) return launch_pp<F16Traits, 32>(a, stream);
    }
    return -1;
}

// Shapes for the SPLIT instances.  AULE_HIP_FWD_PPSPLIT=0 turns the path off (A/B measurements).
bool pp_split_applicable(const FwdArgs& a) {
    static const int on = [] {
        const char* e = getenv("AULE_HIP_FWD_PPSPLIT");
        return (e != nullptr && e[0] == '0') ? 0 : 1;
    }();
    if (!on) return false;
    if (a.dtype != kBF16 && a.dtype != kF16) return false;
    if (a.window > 0) return false;
    // causal: only the bottom-right aligned short chunk (decode / speculative verification / chunked prefill against a
    // KV history): every split then has full work; a top-left mask with Sq < Sk sees only the first Sq keys
    if (a.causal && !(a.coff == a.Sk - a.Sq && a.coff > 0 && a.Sq <= 256)) return false;
    if (a.D != 32 && a.D != 64 && a.D != 128) return false;
    if ((long long)a.Hq / a.Hkv * a.Sq >= (1 << 24)) return false;
    const PPSplitPlan s = pp_split_plan(a);
    const long long tiled_wgs = (long long)a.B * a.Hq * ((a.Sq + kQBlock - 1) / kQBlock);
    // worth it when the plain launch leaves most CUs idle or most waves of a Q block without rows, and the split
    // launch does not: at least two splits, or packing alone folds >= 2 heads into one block
    if (tiled_wgs >= 256 && s.nsplit < 2 && s.nbase * 2 > tiled_wgs) return false;
    if (s.nsplit < 2 && s.nbase == tiled_wgs) return false;   // nothing to gain: same workgroups, plus a combine
    // The path costs a second launch, a stream-ordered allocation and a combine with one workgroup per output row:
    // a floor of ~18 us, and a cost that grows with the rows while the gain grows with Sk.  Measured edges
    // (tools/ppsplit_edges.py, previous behaviour -> this path):
    //   Sk <= 512, B <= 8:  9-16 us -> 18-21 us (loses);  B = 32 (1024 plain workgroups): 62 -> 36 us at Sk 512 (wins)
    //   16 k rows: Sk 1024 27 -> 47 us, Sk 2048 47 -> 54 us (loses); Sk 4096 89 -> 65 us, Sk 8192 166 -> 93 us (wins)
    //   32 k rows: Sk 1024 19 -> 61 us, Sk 4096 67 -> 88 us (loses); Sk 8192 178 -> 165 us (wins)
    // rows <= 4 Sk separates all of these; it is a fit to this sample, not a model.
    if (a.Sk < 1024 && tiled_wgs < 1024) return false;
    if ((long long)a.B * a.Hq * a.Sq > (long long)AULE_PPSPLIT_ROWS_PER_KEY * a.Sk) return false;
    return AULE_PPSPLIT_RULE;
}

int launch_fwd_pp_split(const FwdArgs& a, hipStream_t stream) {
    if (a.dtype == kBF16) {
        if (a.D == 128) return launch_pp_split<Bf16Traits, 128>(a, stream);
        if (a.D == 64) return launch_pp_split<Bf16Traits, 64>(a, stream);
        if (a.D == 32) return launch_pp_split<Bf16Traits, 32>(a, stream);
    } else if (a.dtype == kF16) {
        if (a.D == 128) return launch_pp_split<F16Traits, 128>(a, stream);
        if (a.D == 64) return launch_pp_split<F16Traits, 64>(a, stream);
        if (a.D == 32) return launch_pp_split<F16Traits, 32>(a, stream);
    }
    return -1;
}

int configure_fwd_pp() {
    return set_attr_pp<Bf16Traits, 128>() | set_attr_pp<Bf16Traits, 64>() | set_attr_pp<Bf16Traits, 32>() |
           set_attr_pp<F16Traits, 128>() | set_attr_pp<F16Traits, 64>() | set_attr_pp<F16Traits, 32>();
}

}  // namespace aule_hip
