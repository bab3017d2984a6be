// fa_fwd_ps_gfx950.hip -- FlashAttention-2 forward (16-bit I/O) as a PERSISTENT TILE STREAM.
//
// Same arithmetic, MFMA layouts, LDS images and two-group ping-pong phases as fa_fwd_pp_gfx950.hip (read its header
// and DESIGN.md 3.2 first).  What changes is everything AROUND the tile loop.  The ping-pong kernel runs one
// workgroup per causal Q-block pair and pays, per 256-row Q block, an epilogue (2.5 k cycles), a cold prologue (an
// HBM round trip of ~3.8 k cycles with every CU at a boundary at the same moment), a pre-phase and three alignment
// barriers: 10-14 k cycles in which the matrix pipes idle, against ~65 k cycles of tile work for an average block of
// the headline shape.  Here
//
//   * the grid is one workgroup per CU (LDS allows only one anyway) and every workgroup walks a LIST of Q blocks
//     ("parts"; the (i, n-1-i) pairs of the causal load balance, items g, g+G, g+2G, ... of the launch);
//   * the K/V tile stream does not stop at a Q-block boundary ("seam"): tile positions are numbered through the
//     whole list, the staging cursors (which tile to request next, from which head) simply run on into the next
//     part, so the first tiles of the next block are in LDS when the last tile of this one retires;
//   * at a seam a wave's M-phase is [PV of the last tile | QK^T of the NEXT block's first tile], i.e. the normal
//     M-phase with a different Q: the next block's Q rows are requested into the (dead) Q registers one phase
//     earlier, right after the wave's last QK^T of the block;
//   * a wave's epilogue for a block (normalise, transpose through its LDS slab, whole-row stores, LSE) follows its last PV
//     of the block directly, inside that M-phase: register pressure is lowest there (S and P are dead, the next S is not
//     born yet -- in the next block's first V-phase the same code spilled the K/V staging registers and the Q
//     fragments), and on causal blocks six of the eight waves finish early and do it while they would idle;
//   * no barrier beyond the two per tile step: the two groups stay one phase apart across seams.
//
// The fixed-reference softmax (bf16, and fp16 with a tighter verdict) keeps its per-block range verdict, but a failed block is no longer re-run on the
// spot (that would stall the stream): verdicts are posted per part and, after the stream, the workgroup runs the
// flagged parts again as a second, sparse stream with the online softmax.
//
// Covers: bf16 / fp16, D = 32 / 64 / 128, causal (top-left or shifted by `coff`) and non-causal, any Sq, every part
// with at least 4 KV tiles, no sliding window, no KV split -- everything else stays on fa_fwd_pp_gfx950.hip.
#include <cstdlib>
#include <type_traits>

#include "fa_device.h"
#include "fa_kernels.h"
#include "fa_fwd_tile.h"

namespace aule_hip {
namespace {

struct FwdPSParams {
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
    int pair;     // item = Q blocks (nqb-1-i, i)
    int coff;     // causal position offset (query i sits at position i + coff)
    int nitems;   // nwork * B * Hq; workgroup g takes items g, g + gridDim.x, ...
    // ROPE instances only: rotate Q on its way into the registers (half-split pairs; K arrives rotated).  Tables
    // [rrows, rpitch] fp32, query i uses row i + rpos.
    const float* rcos;
    const float* rsin;
    int rrows, rpitch, rpos;
    // SPLIT instances only (small causal grids, launch_ps_split): a pair of Q blocks is `npiece` work items; a block whose
    // keys are cut into ranges leaves one fp32 partial per range in plane `piece` of `part` ([npiece][B*Hq*Sq][D + 4]:
    // un-normalised O, m in log2 units, l), merged by fa_fwd_ps_combine.
    float* part;
    int part_rows;
    int npiece;
    unsigned magic;   // ps_magic(npiece)
    int pcoff;        // the plan's position offset: coff, or kEverything for a non-causal problem (every key visible to every row)
    unsigned long long* dbg;   // timeline build only: [8 waves][kPSTLMax] tagged s_memtime stamps of workgroup 0
};

constexpr int kPSTLMax = 2048;

// s_setprio levels of the two phases.  Same-box A/B (tools/ps_check.py bench, build/variants): M-phase 0 / V-phase 0 beats
// the predecessor's M-phase 1 by ~1 % on C2 and 3 % on the small shapes; raising the V-phase is within noise of that.
#ifndef AULE_PS_MPRIO
#define AULE_PS_MPRIO 0
#endif
#ifndef AULE_PS_VPRIO
#define AULE_PS_VPRIO 0
#endif
// AULE_PS_YOUNG_PRIO=1: one static s_setprio 1 for the second-dispatched wave group, no per-phase flips (hardware guide, T5
// static form: the younger half loses VALU arbitration to the older one on every segment)
// Operand look-ahead (steps) of the two M-phase loops.  tools/timeline.py with one group idled: the M-phase takes the same
// ~1640 cycles per 32 MFMAs with or without a partner -- it is bound by its own LDS operand latency.
#ifndef AULE_PS_QK_AHEAD
#define AULE_PS_QK_AHEAD 1
#endif
#ifndef AULE_PS_PV_AHEAD
#define AULE_PS_PV_AHEAD 2
#endif
#ifndef AULE_PS_YOUNG_PRIO
#define AULE_PS_YOUNG_PRIO 0
#endif

// AULE_PS_DMA=1: K / V tiles go from global memory straight into LDS (buffer_load ... lds, 1 KiB per wave instruction)
// instead of through 16 staging registers and ds_write_b128 (D >= 64 instances; D = 32 keeps the register path).  A lane
// whose source lies beyond the descriptor's range writes ZEROS into LDS (tools/probe_lds_dma.hip), exactly what the
// register path's out-of-range loads deliver: the ragged last tile needs no clamping here either.
#ifndef AULE_PS_DMA
#define AULE_PS_DMA 1
#endif
#ifndef AULE_PS_DMA_MIN_D
#define AULE_PS_DMA_MIN_D 32
#endif
template <int D> constexpr bool ps_dma() { return AULE_PS_DMA != 0 && D >= AULE_PS_DMA_MIN_D; }
// AULE_PS_DMA_SPREAD=1: a wave's DMA pieces of a step are issued one by one between the four exp blocks of the softmax
// instead of back to back at the start of the V-phase (a piece costs 60-185 issue cycles next to other memory traffic,
// 25-60 in a VALU-only gap: MI355X_MICROARCH "per-instruction cycle constants").  Measured same-box: 7 % SLOWER on every
// D = 128 shape (C2 1007 -> 940 TF), D = 64 unchanged -- the burst at the start of the phase stays.
// table loads in flight per batch of the fused query rotation, in fragment pairs (D = 128 has four pairs)
#ifndef AULE_PS_ROPE_BATCH
#define AULE_PS_ROPE_BATCH 2
#endif
#ifndef AULE_PS_DMA_HOIST
#define AULE_PS_DMA_HOIST 1
#endif
#ifndef AULE_PS_DMA_SPREAD
#define AULE_PS_DMA_SPREAD 0
#endif
// LDS of one workgroup without the part table: register path = Cfg<D>::LDS (2 padded K tiles, 2 V tiles, 8 slabs);
// DMA path = 2 un-padded (swizzled) K tiles, 3 V tiles, 8 slabs.
template <int D> constexpr int ps_tile_lds() {
    return ps_dma<D>() ? 2 * kKVTile * Cfg<D>::RB + 3 * Cfg<D>::VTILE + 8 * Cfg<D>::QSLAB : Cfg<D>::LDS;
}

constexpr int kMaxItems = 64;            // per workgroup (the host sizes the grid accordingly)
constexpr int kMaxSlot = 2 * kMaxItems;  // parts: two per item (the second one invalid for an unpaired block)

__device__ __forceinline__ int rfl(int x) { return __builtin_amdgcn_readfirstlane(x); }

constexpr int kPartPad = 4;   // floats behind the D accumulators of a partial row: m, l, 2 unused (keeps rows 16-byte aligned)

// KV tiles of Q block qb under the causal rule (query i at position i + coff).
__host__ __device__ inline int ps_tiles(int qb, int Sk, int coff) {
    int kv_hi = qb * kQBlock + kQBlock + coff;
    kv_hi = kv_hi < Sk ? kv_hi : Sk;
    kv_hi = kv_hi > 1 ? kv_hi : 1;
    return (kv_hi + kKVTile - 1) / kKVTile;
}
// SPLIT plan of the pair (far, near) of Q blocks: the pair's key tiles, far block's first, are one sequence of
// ntf + ntn tiles cut into n pieces of (nearly) equal length, piece j = [b[j], b[j + 1]) -- at most one range of each
// block.  A cut inside a block stays within the keys EVERY row of the block sees whole (tile index <= first position
// / 64): ranges before it need no mask, and every row of the range behind it sees that range's first key, so all ranges
// run the plain softmax (a finite maximum from their first tile on).  Every range has at least four tiles (the staging
// cursors run three tiles ahead); a cut with no admissible position collapses (b[j] = b[j - 1]: an empty piece).
constexpr int kMaxPieces = 8;
constexpr int kEverything = 1 << 30;   // ps_tiles / ps_cuts position offset of a non-causal problem
constexpr int kSplitMinTiles = 16;   // shortest piece worth a workgroup of its own (ps_split_plan)
struct PSPair {
    int ntf, ntn;
    int b[kMaxPieces + 1];
};
__host__ __device__ inline unsigned ps_magic(int n) { return (unsigned)((0x100000000ull + (unsigned)n - 1) / (unsigned)n); }
// (magic = ps_magic(n), computed on the host: T / n as a multiply-high, exact for T < 2^29 -- a division would drag the
// whole plan from the scalar unit into vector registers)
__host__ __device__ inline PSPair ps_cuts(int far, int near, int Sk, int coff, int n, unsigned magic) {
    PSPair r;
    r.ntf = ps_tiles(far, Sk, coff);
    r.ntn = far != near ? ps_tiles(near, Sk, coff) : 0;
    const int T = r.ntf + r.ntn;
    int fhi = (far * kQBlock + coff) / kKVTile, nhi = (near * kQBlock + coff) / kKVTile;   // last admissible cut inside a block
    fhi = fhi < r.ntf - 4 ? fhi : r.ntf - 4;
    nhi = nhi < r.ntn - 4 ? nhi : r.ntn - 4;
    r.b[0] = 0;
    const int q = (int)(((unsigned long long)(unsigned)T * magic) >> 32), rem = T - q * n;
    // (constant trip counts and indices: on the device the plan stays in registers; one division)
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
__host__ __device__ inline void ps_range(const PSPair& r, int j, int which, int& t0, int& t1) {
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

template <class T, int D, bool CAUSAL, bool RAWOK, bool TL = false, bool ROPE = false, bool SPLIT = false>
__global__ void __launch_bounds__(512, D <= 64 ? 4 : 2) fa_fwd_ps_kernel(const FwdPSParams p) {
    static_assert(!SPLIT || (!TL && !ROPE), "SPLIT instances: no timeline, no fused rotation");
    using C = Cfg<D>;
    using v8 = typename T::v8;
    using std::integral_constant;
    constexpr int RB = C::RB, RBP = C::RBP, CPR = C::CPR, KTILE = C::KTILE, VTILE = C::VTILE;
    constexpr int CH = C::CH, KS = C::KS, DB = C::DB;

    constexpr bool DMA = ps_dma<D>();
    constexpr int KT = DMA ? kKVTile * RB : KTILE;   // bytes of a K tile in LDS (DMA: rows un-padded, 16-byte chunks swizzled)
    constexpr int NVB = DMA ? 3 : 2;                 // V tiles in LDS (DMA: one more -- a tile is requested two phases earlier)
    constexpr int TLDS = ps_tile_lds<D>();
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const Ks = smem;
    char* const Vs = smem + 2 * KT;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = rfl(tid >> 6);
    const int grp = wave >> 2;  // 0: leads, 1: runs one phase behind
    const int l31 = lane & 31, hi = lane >> 5;
    char* const Qs = smem + 2 * KT + NVB * VTILE + wave * C::QSLAB;
    int4* const tab = reinterpret_cast<int4*>(smem + TLDS);                     // [kMaxSlot] {q row offset, kv row offset, qb | -1, -}
    int* const redo = reinterpret_cast<int*>(smem + TLDS + kMaxSlot * 16);      // [kMaxSlot] range verdicts, [kMaxSlot] = any

    const int Sq = p.Sq, Sk = p.Sk, coff = p.coff;
    const float c = p.c;
    int tl_n = 0;
    auto stamp = [&](int tag) __attribute__((always_inline)) {   // timeline build: (tag << 56) | shader clock
        if constexpr (TL) {
            if (blockIdx.x == 0 && tl_n < kPSTLMax) {
                const unsigned long long t = __builtin_amdgcn_s_memtime();
                if (lane == 0) p.dbg[wave * kPSTLMax + tl_n] = (t & 0x00ffffffffffffffull) | ((unsigned long long)tag << 56);
                ++tl_n;
            }
        }
    };

    // ---- part table: thread t describes part (t & 1) of this workgroup's item t >> 1
    const int G = (int)gridDim.x;
    const int nit = (p.nitems - (int)blockIdx.x + G - 1) / G;
    const int nslot = 2 * nit;
    if (tid < nslot) {
        int qb = -1, range = 0;
        if constexpr (SPLIT) {
            // item = (pair, piece): slot 0 its range of the far block, slot 1 of the near block.
            // .z = qb | partial plane + 1 << 24 (0: the range is the whole block -- its O is final), .w = first tile | end tile << 16
            const WorkItem w = decode_work((int)blockIdx.x + (tid >> 1) * G, p.B, p.Hq, p.Hkv, p.npiece * p.nwork, false);
            // (non-causal: no pairing -- the "pair" is one block, far == near)
            const int near = w.blk / p.npiece, piece = w.blk % p.npiece, far = p.pair ? p.nqb - 1 - near : near;
            const PSPair pr = ps_cuts(far, near, Sk, p.pcoff, p.npiece, p.magic);
            int t0, t1;
            ps_range(pr, piece, tid & 1, t0, t1);
            if (t1 > t0) {
                const int whole = (tid & 1) ? pr.ntn : pr.ntf;
                qb = ((tid & 1) ? near : far) | ((t0 == 0 && t1 == whole) ? 0 : (piece + 1) << 24);
                range = t0 | (t1 << 16);
            }
            tab[tid] = int4{(w.b * p.Hq + w.h) * Sq, (w.b * p.Hkv + w.hk) * Sk, qb, range};
        } else {
            const WorkItem w = decode_work((int)blockIdx.x + (tid >> 1) * G, p.B, p.Hq, p.Hkv, p.nwork, false);
            if (p.pair) {
                const int far = p.nqb - 1 - w.blk;       // the larger block of the pair goes first
                if ((tid & 1) == 0) qb = far;
                else if (far != w.blk) qb = w.blk;
            } else if ((tid & 1) == 0) {
                qb = w.blk;
            }
            tab[tid] = int4{(w.b * p.Hq + w.h) * Sq, (w.b * p.Hkv + w.hk) * Sk, qb, 0};
        }
        redo[tid] = 0;
    }
    if (tid == 0) redo[kMaxSlot] = 0;
    __syncthreads();

    auto next_valid = [&](int slot) __attribute__((always_inline)) {
        do {
            ++slot;
        } while (slot < nslot && rfl(tab[slot].z) < 0);
        return slot;
    };
    auto nt_of = [&](int qb) __attribute__((always_inline)) {
        const int kv_hi = CAUSAL ? max(1, min(Sk, qb * kQBlock + kQBlock + coff)) : Sk;
        return (kv_hi + kKVTile - 1) / kKVTile;
    };
    // table entry -> Q block, first KV tile and tile count of the part, partial plane + 1 (0: the part's O is final)
    auto e_qb = [&](const int4& e) __attribute__((always_inline)) { return SPLIT ? (rfl(e.z) & 0xffffff) : rfl(e.z); };
    auto e_t0 = [&](const int4& e) __attribute__((always_inline)) { return SPLIT ? (rfl(e.w) & 0xffff) : 0; };
    auto e_nt = [&](const int4& e) __attribute__((always_inline)) {
        if constexpr (SPLIT) return (int)((unsigned)rfl(e.w) >> 16) - (rfl(e.w) & 0xffff);
        else return nt_of(rfl(e.z));
    };
    auto e_pid = [&](const int4& e) __attribute__((always_inline)) { return SPLIT ? (rfl(e.z) >> 24) : 0; };
    auto head_srd = [&](const void* base, int rowoff, int rows) __attribute__((always_inline)) {
        return make_srd(reinterpret_cast<const char*>(base) + (size_t)(unsigned)rowoff * RB, (unsigned)rows * RB);
    };

    // ---- staging maps (as fa_fwd_pp_gfx950.hip)
    int k_g[CH], k_lds[CH], v_g[CH];
#pragma unroll
    for (int i = 0; i < CH; ++i) {
        const int cidx = tid + 512 * i;
        const int row = cidx / CPR, cc = cidx % CPR;
        k_g[i] = row * RB + cc * 16;
        k_lds[i] = row * RBP + cc * 16;
        const int bidx = (tid >> 3) + 64 * i;
        v_g[i] = ((bidx / (D / 16)) * 4 + ((tid >> 1) & 3)) * RB + ((bidx % (D / 16)) * 2 + (tid & 1)) * 16;
    }
    // DMA: a wave instruction writes 64 x 16 bytes at a wave-uniform LDS address + lane * 16, from per-lane global
    // addresses.  K: piece j = rows of 1 KiB of the un-padded tile; position (row r, chunk c') holds global chunk
    // c' ^ swz(r), swz(r) = (r * CPR / 16) & (CPR - 1) -- with that the 16 lanes ds_read_b128 serves per LDS cycle (rows
    // {0-3, 12-15, 20-27} of one chunk index) hit 16 different 16-byte bank groups.  V: the register path's image is
    // already lane-linear (thread t, chunk i at t * 16 + i * 8192).  Wave w of a group takes pieces 4 i + (w & 3).
    constexpr int KP = DMA ? (kKVTile * RB) / 4096 : 1, VP = DMA ? VTILE / 4096 : 1;
    // Per-lane source offset of piece 0; piece i is 4096 bytes further on in both maps (16 more rows of K; V: the sub-tile
    // index advances by 32), which goes into the instruction's scalar offset.  Recomputed once per step from an opaque
    // lane id: held in a register across the tile loop it gets spilled, and the reload's vmcnt(0) would put the request
    // behind the previous one's HBM round trip.
    auto kdma_off0 = [&]() __attribute__((always_inline)) {
        constexpr int SH = CPR == 16 ? 0 : (CPR == 8 ? 1 : 2);
        int lane_o = lane;
        asm volatile("" : "+v"(lane_o));
        const int q = (wave & 3) * 64 + lane_o, r = q / CPR, cs = q % CPR;
        return r * RB + (cs ^ ((r >> SH) & (CPR - 1))) * 16;
    };
    auto vdma_off0 = [&]() __attribute__((always_inline)) {
        int lane_o = lane;
        asm volatile("" : "+v"(lane_o));
        const int t = (wave & 3) * 64 + lane_o, bidx = t >> 3;
        return ((bidx / (D / 16)) * 4 + ((t >> 1) & 3)) * RB + ((bidx % (D / 16)) * 2 + (t & 1)) * 16;
    };
    static_assert(!DMA || (4 * 64 / CPR) * RB == 4096, "a wave's next piece: 4096 bytes further in the K tile");
    // AULE_PS_DMA_HOIST=1: the offset of the one map a wave uses in the steady state (group 0: V, group 1: K) in a register
    const int dma_off_h = (AULE_PS_DMA_HOIST && DMA) ? (grp == 0 ? vdma_off0() : kdma_off0()) : 0;
    (void)KP; (void)VP; (void)dma_off_h;   // (used in the device pass only)
    constexpr int SWSH = CPR == 16 ? 0 : (CPR == 8 ? 1 : 2);
    const int ka_base = DMA ? l31 * RB + ((((l31 >> SWSH) & (CPR - 1)) ^ hi) * 16) : l31 * RBP + hi * 16;
    const int va_off = hi * (D / 16) * 128 + ((lane >> 4) & 1) * 128 + (lane & 15) * 8;

    u32x4_t kst[CH], vst[CH];
    auto write_k = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < CH; ++i)
            if (C::kFull || tid + 512 * i < C::NCHUNK)
                *reinterpret_cast<u32x4_t*>(Ks + buf * KTILE + k_lds[i]) = kst[i];
    };
    auto write_v = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < CH; ++i)
            if (C::kFull || tid + 512 * i < C::NCHUNK)
                *reinterpret_cast<u32x4_t*>(Vs + buf * VTILE + tid * 16 + i * 8192) = vst[i];
    };

    auto run_stream = [&](auto raw_tag) __attribute__((always_inline)) {
        constexpr bool RAW = decltype(raw_tag)::value != 0;
        int cs = next_valid(-1);   // compute cursor: the part being computed
        if (cs >= nslot) return;

        // ---- staging cursors: the tile each of them requests next (slot, tile in the part, tiles of the part, head)
        // (ks_t / vs_t count from the part's first tile: SPLIT parts start at tile ks_b / vs_b of their head)
        int ks_slot = cs, ks_t = 0, ks_nt, ks_b, vs_slot = cs, vs_t = 0, vs_nt, vs_b;
        __amdgpu_buffer_rsrc_t krs, vrs;
        {
            const int4 e = tab[cs];
            ks_nt = vs_nt = e_nt(e);
            ks_b = vs_b = e_t0(e);
            krs = head_srd(p.k, rfl(e.y), Sk);
            vrs = head_srd(p.v, rfl(e.y), Sk);
        }
        auto adv_k = [&]() __attribute__((always_inline)) {
            if (++ks_t < ks_nt) return;
            ks_slot = next_valid(ks_slot);
            ks_t = 0;
            if (ks_slot < nslot) {
                const int4 e = tab[ks_slot];
                ks_nt = e_nt(e);
                ks_b = e_t0(e);
                krs = head_srd(p.k, rfl(e.y), Sk);
            }
        };
        auto adv_v = [&]() __attribute__((always_inline)) {
            if (++vs_t < vs_nt) return;
            vs_slot = next_valid(vs_slot);
            vs_t = 0;
            if (vs_slot < nslot) {
                const int4 e = tab[vs_slot];
                vs_nt = e_nt(e);
                vs_b = e_t0(e);
                vrs = head_srd(p.v, rfl(e.y), Sk);
            }
        };
        auto issue_k = [&]() __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < CH; ++i)
                if (C::kFull || tid + 512 * i < C::NCHUNK)
                    kst[i] = __builtin_amdgcn_raw_buffer_load_b128(krs, k_g[i], ((SPLIT ? ks_b : 0) + ks_t) * (kKVTile * RB), 0);
        };
        auto issue_v = [&]() __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < CH; ++i)
                if (C::kFull || tid + 512 * i < C::NCHUNK)
                    vst[i] = __builtin_amdgcn_raw_buffer_load_b128(vrs, v_g[i], ((SPLIT ? vs_b : 0) + vs_t) * (kKVTile * RB), 0);
        };
        auto dma_k = [&](int buf) __attribute__((always_inline)) {   // the K tile at the cursor -> Ks[buf] (this wave's pieces)
#if defined(__HIP_DEVICE_COMPILE__)   // (the LDS address-space cast does not exist in the host pass)
            using lds_ptr = __attribute__((address_space(3))) void*;
            const int off0 = AULE_PS_DMA_HOIST ? dma_off_h : kdma_off0();
#pragma unroll
            for (int i = 0; i < KP; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(krs, (lds_ptr)(Ks + buf * KT + (4 * i + (wave & 3)) * 1024), 16, off0,
                                                         ((SPLIT ? ks_b : 0) + ks_t) * (kKVTile * RB) + i * 4096, 0, 0);
#endif
        };
        // one piece of this wave's share of the step's request (group 0: V tile, group 1: K tile), cursors not yet advanced
        auto dma_piece = [&](int i, int P) __attribute__((always_inline)) {
#if defined(__HIP_DEVICE_COMPILE__)
            using lds_ptr = __attribute__((address_space(3))) void*;
            if (grp == 0) {
                if (i < VP && vs_slot < nslot)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(vrs, (lds_ptr)(Vs + ((P + 1) % 3) * VTILE + (4 * i + (wave & 3)) * 1024), 16,
                                                             vdma_off0(), ((SPLIT ? vs_b : 0) + vs_t) * (kKVTile * RB) + i * 4096, 0, 0);
            } else {
                if (i < KP && ks_slot < nslot)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(krs, (lds_ptr)(Ks + (P & 1) * KT + (4 * i + (wave & 3)) * 1024), 16,
                                                             kdma_off0(), ((SPLIT ? ks_b : 0) + ks_t) * (kKVTile * RB) + i * 4096, 0, 0);
            }
#endif
        };
        auto dma_v = [&](int buf) __attribute__((always_inline)) {
#if defined(__HIP_DEVICE_COMPILE__)
            using lds_ptr = __attribute__((address_space(3))) void*;
            const int off0 = AULE_PS_DMA_HOIST ? dma_off_h : vdma_off0();
#pragma unroll
            for (int i = 0; i < VP; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(vrs, (lds_ptr)(Vs + buf * VTILE + (4 * i + (wave & 3)) * 1024), 16, off0,
                                                         ((SPLIT ? vs_b : 0) + vs_t) * (kKVTile * RB) + i * 4096, 0, 0);
#endif
        };
        // phase barriers.  DMA: raw s_barrier (a fence would drain the DMA queue at every barrier); a group waits for the
        // pieces it requested at the start of its V-phase at the end of the M-phase that follows (one tile step of flight).
        auto phase_barrier = [&](bool end_of_m) __attribute__((always_inline)) {
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (DMA) {
                if (end_of_m) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                else asm volatile("" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
            } else {
                __syncthreads();
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        bool have_k = true, have_v = true;   // kst / vst hold a requested tile that is not in LDS yet

        // ---- compute-side state
        f32x16_t o[DB];
        f32x16_t s[2];
        v8 pb[2][2];
        u32x4_t qx[KS];   // Q fragments (B operand of S^T = K.Q^T): lane (q, hi) holds d = 16ks+8hi..+7.  ONE variable for the
                          // fragments in use and the next part's in flight: their lifetimes are disjoint (see tile_step)
        float m = 0.f, l = 0.f;
        const unsigned flip = p.negq ? 0x80008000u : 0u;

        auto issue_q = [&](int qoff, int q0) __attribute__((always_inline)) {   // rows >= Sq read as 0
            const __amdgpu_buffer_rsrc_t qrs = head_srd(p.q, qoff, Sq);
            // (opaque lane id: a row offset hoisted to kernel entry gets spilled, and the reload's vmcnt(0) would sit
            // between this step's K/V requests and these loads)
            int lane_o = lane;
            asm volatile("" : "+v"(lane_o));
            const int qoffs = (q0 + (lane_o & 31)) * RB + (lane_o >> 5) * 16;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
                qx[ks] = __builtin_amdgcn_raw_buffer_load_b128(qrs, qoffs + ks * 32, 0, 0);
        };
        // ROPE: the rotation of rope_gfx950.hip (same fp32 expression, same single rounding: the fragments come out bit-identical
        // to a separate pass over Q), on the registers the Q request landed in.  Half-split pairs (d, d + D/2) are the
        // fragments ks and ks + KS/2 of the same lane.  Table rows beyond the table read as 0 (descriptor bounds); they belong
        // to rows >= Sq, whose Q is 0 already.  D = 128: two batches of 32 table registers (the accumulators are live).
        auto rotate_q = [&](int q0) __attribute__((always_inline)) {
            if constexpr (ROPE) {
                const unsigned tbytes = (unsigned)p.rrows * (unsigned)p.rpitch * 4u;
                const __amdgpu_buffer_rsrc_t crs = make_srd(p.rcos, tbytes), srs = make_srd(p.rsin, tbytes);
                int lane_o = lane;
                asm volatile("" : "+v"(lane_o));
                const unsigned toff = (unsigned)(q0 + (lane_o & 31) + p.rpos) * (unsigned)(p.rpitch * 4) + (unsigned)((lane_o >> 5) * 32);   // (< 2^32: fwd_ps_rope_fusable)
                constexpr int HK = KS / 2, BATCH = HK < AULE_PS_ROPE_BATCH ? HK : AULE_PS_ROPE_BATCH;
#pragma unroll
                for (int k0 = 0; k0 < HK; k0 += BATCH) {
                    u32x4_t tc[BATCH][2], ts[BATCH][2];
#pragma unroll
                    for (int b = 0; b < BATCH; ++b)
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            tc[b][h] = __builtin_amdgcn_raw_buffer_load_b128(crs, toff + (k0 + b) * 64 + h * 16, 0, 0);
                            ts[b][h] = __builtin_amdgcn_raw_buffer_load_b128(srs, toff + (k0 + b) * 64 + h * 16, 0, 0);
                        }
#pragma unroll
                    for (int b = 0; b < BATCH; ++b)
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const unsigned ua = qx[k0 + b][j], ub = qx[k0 + b + HK][j];
                            // (whole-vector casts: __builtin_bit_cast applied to a vector ELEMENT reads element 0 whatever the index)
                            const f32x4_t cv = __builtin_bit_cast(f32x4_t, tc[b][j >> 1]), sv = __builtin_bit_cast(f32x4_t, ts[b][j >> 1]);
                            const float c0 = cv[(j & 1) * 2], c1 = cv[(j & 1) * 2 + 1];
                            const float s0 = sv[(j & 1) * 2], s1 = sv[(j & 1) * 2 + 1];
                            float y1l, y2l, y1h, y2h;
                            rope_pair(T::lo(ua), T::lo(ub), c0, s0, y1l, y2l);
                            rope_pair(T::hi(ua), T::hi(ub), c1, s1, y1h, y2h);
                            qx[k0 + b][j] = T::pack2(y1l, y1h);
                            qx[k0 + b + HK][j] = T::pack2(y2l, y2h);
                        }
                }
            }
        };
        auto take_q = [&]() __attribute__((always_inline)) {   // negative scale: flip the sign of Q once, in place
            if (flip != 0u) {
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    qx[ks][0] ^= flip; qx[ks][1] ^= flip; qx[ks][2] ^= flip; qx[ks][3] ^= flip;
                }
            }
        };

        auto qk = [&](int buf) __attribute__((always_inline)) {  // S^T = K_tile . Q^T
            const char* kb = Ks + buf * KT + (DMA ? 0 : ka_base);
            constexpr int kAhead = AULE_PS_QK_AHEAD;
            u32x4_t kf[KS][2];
            auto rd = [&](int ks) __attribute__((always_inline)) {
                if constexpr (DMA) {   // swizzled chunk: (2 ks + hi) ^ swz(row) = one XOR on the byte offset
                    const int a = ka_base ^ (ks * 32);
                    kf[ks][0] = *reinterpret_cast<const u32x4_t*>(kb + a);
                    kf[ks][1] = *reinterpret_cast<const u32x4_t*>(kb + a + 32 * RB);
                } else {
                    kf[ks][0] = *reinterpret_cast<const u32x4_t*>(kb + ks * 32);
                    kf[ks][1] = *reinterpret_cast<const u32x4_t*>(kb + ks * 32 + 32 * RBP);
                }
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
                s[0] = T::mfma(as_v8<T>(kf[ks][0]), as_v8<T>(qx[ks]), ks == 0 ? z : s[0]);
                s[1] = T::mfma(as_v8<T>(kf[ks][1]), as_v8<T>(qx[ks]), ks == 0 ? z : s[1]);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if (ks + kAhead < KS) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            }
        };
        auto pv = [&](int buf) __attribute__((always_inline)) {  // O^T += V^T . P^T
            const char* vb = Vs + buf * VTILE + va_off;
            constexpr int NST = 4 * DB;
            constexpr int kAhead = AULE_PS_PV_AHEAD;
            s16x4_t a0[NST], a1[NST];
            auto rd = [&](int st) __attribute__((always_inline)) {
                const int sk = st / DB, d = st % DB;
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

        // ---- part scalars of the compute side
        int qoff, qb, nt, na, q0w, qposv, P0 = 0;
        int tb = 0, pid = 0;                       // SPLIT: first KV tile of the part, partial plane + 1
        int n_slot, n_qoff = 0, n_qb = 0;          // the part after this one (n_slot == nslot: none)
        auto enter_part = [&](int slot) __attribute__((always_inline)) {
            const int4 e = tab[slot];
            qoff = rfl(e.x);
            qb = e_qb(e);
            nt = e_nt(e);
            q0w = qb * kQBlock + wave * 32;
            qposv = q0w + l31 + coff;
            const int wave_kv_hi = CAUSAL ? min(Sk, q0w + 32 + coff) : Sk;
            na = max(1, (wave_kv_hi + kKVTile - 1) / kKVTile);
            if constexpr (SPLIT) {
                tb = e_t0(e);
                pid = e_pid(e);
                na = max(1, min(na - tb, nt));     // the wave's active tiles INSIDE the part's range
            }
            n_slot = next_valid(slot);
            if (n_slot < nslot) {
                const int4 en = tab[n_slot];
                n_qoff = rfl(en.x);
                n_qb = e_qb(en);
            }
        };

        // ---- epilogue of a part, in two pieces.
        // (a) epilogue_pack, in the M-phase right behind the wave's last PV of the part: O = O^T / l, rounded and written
        //     transposed into the wave's slab; LSE; range verdict.  Frees the 64 accumulators for the next part.
        // (b) drain_rows, two 16-byte row chunks per lane in each of the following V-phases (D = 128: four steps): slab
        //     -> registers -> whole-row global stores.  Spreading them keeps (a) short -- all eight stores at once cost
        //     2-4 k cycles inside an M-phase the partner group waits for -- and keeps the store queue shallow.
        //     Buffer stores against a descriptor of the head's Sq rows: rows >= Sq are dropped by the bounds check.
        // No division, no LDS-based lane exchange (this runs beside the partner's LDS traffic).
        constexpr int NR = (32 * CPR) / 64;      // row chunks per lane (CPR / 2)
        constexpr int kDrain = NR >= 2 ? 2 : 1;  // per V-phase
        int ep_left = 0, ep_qoff = 0, ep_q0w = 0;
        auto drain_rows = [&](int n) __attribute__((always_inline)) {
            int lane_o = lane;   // opaque copy: keeps the slab / row addresses from being hoisted out of the part loop and spilled
            asm volatile("" : "+v"(lane_o));
            const __amdgpu_buffer_rsrc_t ors = head_srd(p.o, ep_qoff, Sq);
            const int i0 = NR - ep_left;                                  // first chunk of this call
            const int row0 = lane_o / CPR + (64 / CPR) * i0, cc = lane_o % CPR;   // chunk i covers rows lane / CPR + (64 / CPR) i
            const char* src = Qs + row0 * RBP + cc * 16;
            const int dst = (ep_q0w + row0) * RB + cc * 16;
            u32x4_t x[kDrain];
#pragma unroll
            for (int i = 0; i < kDrain; ++i)
                if (i < n) x[i] = *reinterpret_cast<const u32x4_t*>(src + i * (64 / CPR) * RBP);
#pragma unroll
            for (int i = 0; i < kDrain; ++i)
                if (i < n) __builtin_amdgcn_raw_buffer_store_b128(x[i], ors, dst + i * (64 / CPR) * RB, 0, 0);
            ep_left -= n;
        };
        auto epilogue_pack = [&](int eqoff, int eq0w) __attribute__((always_inline)) {
            while (ep_left > 0) drain_rows(ep_left < kDrain ? ep_left : kDrain);   // (only when two epilogues come < NR / 2 steps apart)
            int lane_o = lane;
            asm volatile("" : "+v"(lane_o));
            const int l31 = lane_o & 31, hi = lane_o >> 5;
            stamp(0xd0);
            const float lt = l + xhalf_fast(l);
            float inv = __builtin_amdgcn_rcpf(lt);   // 1 ulp; O is rounded to 8 / 11 bits right after
            if constexpr (TL) { asm volatile("s_nop 0" : "+v"(inv)); stamp(0xd1); }
#pragma unroll
            for (int d = 0; d < DB; ++d)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    u32x2_t u;
                    u[0] = T::pack2(o[d][4 * g4 + 0] * inv, o[d][4 * g4 + 1] * inv);
                    u[1] = T::pack2(o[d][4 * g4 + 2] * inv, o[d][4 * g4 + 3] * inv);
                    *reinterpret_cast<u32x2_t*>(Qs + l31 * RBP + (32 * d + 8 * g4 + 4 * hi) * 2) = u;
                }
            if constexpr (TL) { __builtin_amdgcn_s_waitcnt(0xc07f); stamp(0xd2); }
            {   // LSE: lanes of the upper half and a null pointer fall outside the descriptor
                const __amdgpu_buffer_rsrc_t lrs = make_srd(p.lse + (size_t)(unsigned)eqoff, p.lse != nullptr ? (unsigned)Sq * 4u : 0u);
                const float lse = (m + fast_log2(lt)) * kLn2;
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, lse), lrs, hi == 0 ? (eq0w + l31) * 4 : 0x7ffffff0, 0, 0);
            }
            if constexpr (RAW) {   // range verdict of the fixed-reference pass (NaN fails it too)
                // bf16 weights have the fp32 exponent range: the row sum only has to stay finite.  fp16 weights overflow at
                // 65504: a row sum below 2^15 proves that no single weight reached it (weights are positive), i.e. that no
                // logit exceeded the first tile's maximum by 15 in log2 units; keys so far below it that their fp16 weight
                // underflows carry < 2^-24 of the first tile's maximum weight.  (A row of > 2^15 near-equal logits fails
                // the verdict without having overflowed and is merely recomputed with the online form.)
                const bool ok = (lt > 0x1p-100f) && (lt < (T::kDType == 2 ? 0x1p110f : 0x1p15f));
                if (__builtin_amdgcn_ballot_w64(!ok) != 0 && lane == 0) redo[cs] = redo[kMaxSlot] = 1;
            }
            ep_left = NR; ep_qoff = eqoff; ep_q0w = eq0w;
            stamp(0xd4);
        };

        // SPLIT: a part that covers only a range of its block's keys leaves the un-normalised accumulators, m and l as an
        // fp32 partial row (16-byte stores straight from the accumulator layout: lane (q, hi) owns columns
        // 32 d + 8 g + 4 hi .. + 3 of row q); fa_fwd_ps_combine merges the two planes.  Rows >= Sq fall outside the descriptor.
        auto partial_store = [&](int eqoff, int eq0w, int plane) __attribute__((always_inline)) {
            if constexpr (SPLIT) {
                constexpr int PP = (D + kPartPad) * 4;   // bytes per partial row
                int lane_o = lane;
                asm volatile("" : "+v"(lane_o));
                const int l31 = lane_o & 31, hi = lane_o >> 5;
                const float lt = l + xhalf_fast(l);
                float* const base = p.part + ((size_t)plane * (size_t)(unsigned)p.part_rows + (size_t)(unsigned)eqoff) * (D + kPartPad);
                const __amdgpu_buffer_rsrc_t prs = make_srd(base, (unsigned)Sq * (unsigned)PP);
                const int roff = (eq0w + l31) * PP;
#pragma unroll
                for (int d = 0; d < DB; ++d)
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        const f32x4_t x = {o[d][4 * g4 + 0], o[d][4 * g4 + 1], o[d][4 * g4 + 2], o[d][4 * g4 + 3]};
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, x), prs, roff + (32 * d + 8 * g4 + 4 * hi) * 4, 0, 0);
                    }
                const u32x2_t ml = {__builtin_bit_cast(unsigned, m), __builtin_bit_cast(unsigned, lt)};
                __builtin_amdgcn_raw_buffer_store_b64(ml, prs, hi == 0 ? roff + D * 4 : 0x7ffffff0, 0, 0);
                if constexpr (RAW) {
                    const bool ok = (lt > 0x1p-100f) && (lt < (T::kDType == 2 ? 0x1p110f : 0x1p15f));
                    if (__builtin_amdgcn_ballot_w64(!ok) != 0 && lane == 0) redo[cs] = redo[kMaxSlot] = 1;
                }
            }
        };

        auto softmax = [&](int kv0, auto sm_tag, auto&& gap) __attribute__((always_inline)) {   // gap(i): filler before exp block i
            constexpr int SM = decltype(sm_tag)::value;   // 0 online (lazy rescale), 1 fixed reference, 2 first tile of a part
            const bool need_mask = (CAUSAL && (kv0 + kKVTile - 1 > q0w + coff)) || (kv0 + kKVTile > Sk);
            if (need_mask) {
                // (opaque copy: without it LICM hoists the 32 per-lane key indices and compare masks out of the part
                // loop and spills them around every seam)
                int hi_o = hi;
                asm volatile("" : "+v"(hi_o));
#pragma unroll
                for (int sb = 0; sb < 2; ++sb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int kv = kv0 + sb * 32 + crow(r, hi_o);
                        const bool vis = (kv < Sk) && (!CAUSAL || kv <= qposv);
                        s[sb][r] = vis ? s[sb][r] : -INFINITY;
                    }
            }
            if constexpr (SM != 1) {
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
                if constexpr (SM == 2) {
                    m = mxc;   // first tile: O = 0 (zeroed behind the previous epilogue), nothing to rescale (key 0 is visible
                    l = 0.f;   // to every row: mxc is finite)
                } else {
                    if (__builtin_amdgcn_ballot_w64(mxc > m + kRescaleThr) != 0) {
                        const float m_new = fmaxf(m, mxc);
                        const float alpha = fast_exp2(m - m_new);
                        m = m_new;
                        l *= alpha;
#pragma unroll
                        for (int d = 0; d < DB; ++d)
#pragma unroll
                            for (int r = 0; r < 16; ++r) o[d][r] *= alpha;
                    }
                }
            }
            const float nm = -m;
            float a0 = 0.f, a1 = 0.f;
            u32x4_t pr[2][2];
#pragma unroll
            for (int sb = 0; sb < 2; ++sb)
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    gap(2 * sb + kk);
                    pr[sb][kk] = softmax_oct<T>(s[sb][8 * kk], s[sb][8 * kk + 1], s[sb][8 * kk + 2], s[sb][8 * kk + 3],
                                                  s[sb][8 * kk + 4], s[sb][8 * kk + 5], s[sb][8 * kk + 6], s[sb][8 * kk + 7],
                                                  c, nm, a0, a1);
                }
            l += a0 + a1;
            // pin the results of this phase HERE (register-only code is otherwise sunk past the barrier)
            asm volatile("" : "+v"(pr[0][0]), "+v"(pr[0][1]), "+v"(pr[1][0]), "+v"(pr[1][1]), "+v"(l), "+v"(m));
#pragma unroll
            for (int sb = 0; sb < 2; ++sb)
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) pb[sb][kk] = as_v8<T>(pr[sb][kk]);
        };

        // ---- one tile step = V-phase + barrier + M-phase + barrier, for stream position P (tile j = P - P0 of the part).
        //      MODE 2 = softmax, PV, QK^T of the next tile; 1 = softmax, PV (the wave's last active tile of the part);
        //      0 = tile fully masked for this wave (staging and barriers only).  SM as in softmax().  FIRST: first tile
        //      of a part (finish the previous part's O first).  TAIL steps (MODE <= 1) also carry the seam duties:
        //      request the next part's Q after the wave's last QK^T, and compute the next part's S_0 at its last tile.
        auto tile_step = [&](int P, auto mode_tag, auto sm_tag, auto first_tag) __attribute__((always_inline)) {
            constexpr int MODE = decltype(mode_tag)::value;
            constexpr bool FIRST = decltype(first_tag)::value != 0;
            const int j = P - P0;
            const int tlt = 8 * MODE + (FIRST ? 32 : 0) + ((MODE <= 1 && j == nt - 1 && n_slot < nslot) ? 64 : 0);
            stamp(tlt + 1);
            // ---- V-phase: staging.  Group d writes the tiles it requested one step ago (V of position P + d,
            //      K of position P + 1 + d) and requests the next ones (hazards: DESIGN.md "forward schedule";
            //      positions run through the seams, so nothing changes there).
            if constexpr (DMA) {
                // V cursor at tile P + 1, K cursor at tile P + 2 (every wave keeps both; group 0 requests V, group 1 K.  Letting a
                // wave advance only its own cursor cost 10 %: the two branch bodies each keep a descriptor copy live, and the
                // scalar registers spill).
                // V_{P+1} -> Vs[(P+1) % 3]: its last reader (PV of tile P - 2) finished two phases ago.  K_{P+2} -> Ks[P & 1]:
                // its last reader (QK^T of tile P, group 1's M-phase(P - 1)) finished in the phase before this one.
                have_v = vs_slot < nslot;
                have_k = ks_slot < nslot;
                if constexpr (!(AULE_PS_DMA_SPREAD != 0 && MODE >= 1)) {
                    if (grp == 0) {
                        if (have_v) dma_v((P + 1) % 3);
                    } else {
                        if (have_k) dma_k(P & 1);
                    }
                    if (have_v) adv_v();
                    if (have_k) adv_k();
                }
            } else {
#ifndef AULE_PS_X_NOWRITE   // (timing experiments only, tools/ps_experiments.sh: results are garbage with either flag)
                if (have_v) write_v((P + grp) & 1);
                if (have_k) write_k((P + 1 + grp) & 1);
#endif
                have_v = vs_slot < nslot;
                have_k = ks_slot < nslot;
#ifdef AULE_PS_X_NOLOAD
                if (have_v) adv_v();
                if (have_k) adv_k();
#else
                if (have_v) { issue_v(); adv_v(); }
                if (have_k) { issue_k(); adv_k(); }
#endif
            }
            if (ep_left > 0) drain_rows(kDrain);
            stamp(tlt + 2);
            if constexpr (MODE == 1) {
                // the wave's last QK^T of this part is behind it (M-phase of step na - 2): the Q registers are free
                if (n_slot < nslot) issue_q(n_qoff, n_qb * kQBlock + wave * 32);
            }
            stamp(tlt + 3);
            if constexpr (!AULE_PS_YOUNG_PRIO) __builtin_amdgcn_s_setprio(AULE_PS_VPRIO);
            if constexpr (MODE >= 1) {
                if constexpr (DMA && AULE_PS_DMA_SPREAD != 0) {
                    softmax(((SPLIT ? tb : 0) + j) * kKVTile, sm_tag, [&](int i) __attribute__((always_inline)) { dma_piece(i, P); });
                    if (have_v) adv_v();
                    if (have_k) adv_k();
                } else {
                    softmax(((SPLIT ? tb : 0) + j) * kKVTile, sm_tag, [](int) {});
                }
            }
            if constexpr (!AULE_PS_YOUNG_PRIO) __builtin_amdgcn_s_setprio(0);
            stamp(tlt + 4);
            phase_barrier(false);
            stamp(tlt + 5);
            // ---- M-phase
            if constexpr (!AULE_PS_YOUNG_PRIO) __builtin_amdgcn_s_setprio(AULE_PS_MPRIO);
            if constexpr (MODE >= 1) pv(DMA ? P % 3 : (P & 1));
            if constexpr (TL && MODE >= 1) { keep_live(o[0], o[DB - 1]); stamp(tlt + 6); }
            if constexpr (MODE == 2) {
                __builtin_amdgcn_sched_barrier(0);
                qk((P + 1) & 1);
            }
            if constexpr (MODE == 1) {   // this wave's O of the part is final
                __builtin_amdgcn_sched_barrier(0);
                if (SPLIT && pid != 0) partial_store(qoff, q0w, pid - 1);
                else epilogue_pack(qoff, q0w);
#pragma unroll
                for (int d = 0; d < DB; ++d)
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
            }
            if constexpr (MODE <= 1) {
                if (j == nt - 1 && n_slot < nslot) {   // seam: S_0 of the next part, with its Q
                    __builtin_amdgcn_sched_barrier(0);
                    rotate_q(n_qb * kQBlock + wave * 32);
                    take_q();
                    if constexpr (TL) { asm volatile("s_nop 0" : "+v"(qx[0]), "+v"(qx[KS - 1])); stamp(tlt + 6); }
                    qk((P + 1) & 1);
                }
            }
            if constexpr (TL) { keep_live(s[0], s[1]); stamp(tlt + 7); }
            if constexpr (!AULE_PS_YOUNG_PRIO) __builtin_amdgcn_s_setprio(0);
            phase_barrier(true);
        };

        // ---- prologue of the stream: tiles 0, 1, 2 of the first part (it has >= 4) and its Q, in one HBM round trip.
        //      Entry state of step 0: K_0 in LDS; group 0 holds (V_0, K_1), group 1 has written them and holds (V_1, K_2).
        stamp(0xe0);
        if constexpr (AULE_PS_YOUNG_PRIO != 0) {
            if (grp == 1) __builtin_amdgcn_s_setprio(1);   // (grp comes from readfirstlane: a scalar branch around one s_setprio)
        }
#pragma unroll
        for (int d = 0; d < DB; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
        enter_part(cs);
        if constexpr (DMA) {
            // K_0 -> Ks[0], K_1 -> Ks[1] (group 1's waves), V_0 -> Vs[0] (group 0's): with Q, one HBM round trip.  Leaves
            // the K cursor at tile 2 and the V cursor at tile 1, the state step 0 expects.
            if (grp == 1) dma_k(0); else dma_v(0);
            adv_k();
            adv_v();
            if (grp == 1) dma_k(1);
            adv_k();
            issue_q(qoff, q0w);
            rotate_q(q0w);
            take_q();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (grp == 1) __builtin_amdgcn_s_barrier();  // group 1 starts one phase late
        } else {
            issue_k(); adv_k();      // K_0
            issue_v(); adv_v();      // V_0
            u32x4_t kpre1[CH], vpre1[CH], kpre2[CH];
    #pragma unroll
            for (int i = 0; i < CH; ++i)
                if (C::kFull || tid + 512 * i < C::NCHUNK) {
                    kpre1[i] = __builtin_amdgcn_raw_buffer_load_b128(krs, k_g[i], ((SPLIT ? tb : 0) + 1) * (kKVTile * RB), 0);
                    if (grp == 1) {
                        vpre1[i] = __builtin_amdgcn_raw_buffer_load_b128(vrs, v_g[i], ((SPLIT ? tb : 0) + 1) * (kKVTile * RB), 0);
                        kpre2[i] = __builtin_amdgcn_raw_buffer_load_b128(krs, k_g[i], ((SPLIT ? tb : 0) + 2) * (kKVTile * RB), 0);
                    }
                }
            adv_k();                 // K_1 requested: the K cursor stands at tile 2, the V cursor at tile 1
            if (grp == 1) { adv_k(); adv_v(); }
            issue_q(qoff, q0w);
            rotate_q(q0w);
            take_q();
            write_k(0);
    #pragma unroll
            for (int i = 0; i < CH; ++i) kst[i] = kpre1[i];
            if (grp == 1) {
                write_v(0);
                write_k(1);
    #pragma unroll
                for (int i = 0; i < CH; ++i) {
                    vst[i] = vpre1[i];
                    kst[i] = kpre2[i];
                }
            }
            __syncthreads();
            if (grp == 1) __syncthreads();  // group 1 starts one phase late
        }
        stamp(0xe1);
        qk(0);                          // pre-phase: S_0
        if constexpr (TL) { keep_live(s[0], s[1]); stamp(0xe2); }
        phase_barrier(false);

        // ---- the stream
        for (;;) {
            int P = P0;
            constexpr int SMF = RAW ? 1 : 0;
            if (na > 1) {
                tile_step(P, integral_constant<int, 2>{}, integral_constant<int, 2>{}, integral_constant<int, 1>{});
                for (++P; P + 1 < P0 + na; ++P)
                    tile_step(P, integral_constant<int, 2>{}, integral_constant<int, SMF>{}, integral_constant<int, 0>{});
                tile_step(P, integral_constant<int, 1>{}, integral_constant<int, SMF>{}, integral_constant<int, 0>{});
            } else {
                tile_step(P, integral_constant<int, 1>{}, integral_constant<int, 2>{}, integral_constant<int, 1>{});
            }
            for (++P; P < P0 + nt; ++P)
                tile_step(P, integral_constant<int, 0>{}, integral_constant<int, 0>{}, integral_constant<int, 0>{});
            if (n_slot >= nslot) break;
            P0 += nt;
            cs = n_slot;
            enter_part(cs);
        }
        stamp(0xf0);
        while (ep_left > 0) drain_rows(ep_left < kDrain ? ep_left : kDrain);
        if (grp == 0) phase_barrier(false);  // pairs with group 1's last phase barrier: all waves aligned again
        stamp(0xf1);
    };

    if constexpr (RAWOK) {
        run_stream(std::integral_constant<int, 1>{});
        __syncthreads();   // every verdict posted, every LDS tile buffer idle
        if (rfl(redo[kMaxSlot]) != 0) {
            __syncthreads();
            if (tid < nslot && redo[tid] == 0) tab[tid].z = -1;   // second, sparse stream: only the flagged parts
            __syncthreads();
            run_stream(std::integral_constant<int, 0>{});
        }
    } else {
        run_stream(std::integral_constant<int, 0>{});
    }
}

template <class T, int D, bool RAWOK>
int launch_ps(const FwdArgs& a, hipStream_t stream) {
    FwdPSParams p;
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
    p.coff = a.causal ? a.coff : 0;
    p.nitems = p.nwork * a.B * a.Hq;
    p.dbg = nullptr;
    p.rcos = a.rope_cos; p.rsin = a.rope_sin; p.rrows = a.rope_rows; p.rpitch = a.rope_pitch; p.rpos = a.rope_pos;
    p.part = nullptr; p.part_rows = 0; p.npiece = 0; p.magic = 0; p.pcoff = 0;
    // one workgroup per CU (two for D <= 64); more only when a workgroup's list would not fit its part table
    const long long ncu = (long long)device_cu_count(a.device) * (D <= 64 ? 2 : 1);
    const long long rounds = (p.nitems + ncu * kMaxItems - 1) / (ncu * kMaxItems);
    long long G = ncu * rounds;
    if (G > p.nitems) G = p.nitems;
    const dim3 grid((unsigned)G), block(512);
    const size_t lds = ps_tile_lds<D>() + kMaxSlot * 20 + 16;
    if constexpr (RAWOK && D >= 64) {
        if (a.rope_cos != nullptr) {
            if (a.causal)
                hipLaunchKernelGGL((fa_fwd_ps_kernel<T, D, true, true, false, true>), grid, block, lds, stream, p);
            else
                hipLaunchKernelGGL((fa_fwd_ps_kernel<T, D, false, true, false, true>), grid, block, lds, stream, p);
            return (int)hipGetLastError();
        }
    }
    if (a.rope_cos != nullptr) return -1;
    if (a.causal)
        hipLaunchKernelGGL((fa_fwd_ps_kernel<T, D, true, RAWOK>), grid, block, lds, stream, p);
    else
        hipLaunchKernelGGL((fa_fwd_ps_kernel<T, D, false, RAWOK>), grid, block, lds, stream, p);
    return (int)hipGetLastError();
}

// Merge of the partial planes of every Q block the SPLIT plan cut into ranges: one thread per four columns of a row,
// 1024 / D rows per 256-thread workgroup; blockIdx = (row group, Q block, b * Hq + h).  Blocks the plan left whole were
// finished by the stream kernel: their workgroups exit.  Bound: HBM (ranges x (D + 4) x 4 bytes read, D x 2 + 4 written
// per row).
template <class T, int D, int N>
__global__ void __launch_bounds__(256) fa_fwd_ps_combine(const FwdPSParams p) {
    constexpr int TPR = D / 4, RPW = 256 / TPR, PP = D + kPartPad;
    const int bh = (int)blockIdx.z, qb = (int)blockIdx.y;
    const int mirror = p.pair ? p.nqb - 1 - qb : qb, near = qb < mirror ? qb : mirror, far = p.pair ? p.nqb - 1 - near : near;
    const int which = (qb == far) ? 0 : 1;
    const PSPair pr = ps_cuts(far, near, p.Sk, p.pcoff, p.npiece, p.magic);
    const int row = qb * kQBlock + (int)blockIdx.x * RPW + (int)threadIdx.x / TPR;
    if (row >= p.Sq) return;
    const int c4 = ((int)threadIdx.x % TPR) * 4;
    const size_t grow = (size_t)bh * p.Sq + row;
    // which planes hold a range of this block (uniform over the workgroup: scalar code)
    unsigned mask = 0;
#pragma unroll
    for (int j = 0; j < kMaxPieces; ++j) {
        int t0, t1;
        ps_range(pr, j, which, t0, t1);
        if (j < N && t1 > t0) mask |= 1u << j;
    }
    if (__builtin_popcount(mask) < 2) return;   // one range = the whole block: the stream kernel finished it
    // every plane's loads first (independent: one round trip), then the merge.  N = npiece is a template parameter and the
    // loads are unconditional -- a plane without a range of this block reads the first plane that has one, with weight 0
    // -- because per-plane branches around the loads make the compiler copy the whole register array at every merge point.
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

// Small causal grids: the paired launch has fewer items than the chip has workgroup slots, and a pair cannot be made
// shorter by cutting its query rows (a workgroup's time is its KEY tiles: 128-row blocks would take as long).  So each
// pair becomes n items of 1/n of its key tiles (ps_cuts), a block cut into ranges leaves one fp32 partial per range, one
// more launch merges them.  (measured: tools/ps_split_check.py.)
// AULE_HIP_FWD_PSSPLIT=<n>: at most n pieces per pair; 0 (or 1) turns the path off (A/B measurements)
static int ps_split_max_pieces() {
    static const int v = [] {
        const char* e = getenv("AULE_HIP_FWD_PSSPLIT");
        if (e == nullptr || e[0] < '0' || e[0] > '9') return kMaxPieces;
        const int n = atoi(e);
        return n < kMaxPieces ? n : kMaxPieces;
    }();
    return v;
}

struct PSSplitPlan {
    bool ok;
    int nqb, nwork, n;
    long long nitems;
    size_t bytes;
};
static PSSplitPlan ps_split_plan(const FwdArgs& a, int slots) {
    PSSplitPlan s{};
    const int pcoff = a.causal ? a.coff : kEverything;
    s.nqb = (a.Sq + kQBlock - 1) / kQBlock;
    s.nwork = a.causal ? (s.nqb + 1) / 2 : s.nqb;   // pairs of blocks, or single blocks
    const long long pairs = (long long)s.nwork * a.B * a.Hq;
    // as many pieces as still fit the chip in one round (one workgroup per CU, also at D = 64: a second workgroup on a CU
    // shares its matrix pipes), each at least kSplitMinTiles tiles of the longest pair.  Measured (tools/ps_split_prof.sh):
    // pieces of 17+ tiles win (S4096 H8: 102 -> 71 us with 2 pieces, 59 us with 4), 9-10 tiles are a wash at D = 128
    // (S1024 H32: 39.5 vs 40.1 us) and a loss at D = 64 (S2048 H32, 4 pieces: 57 -> 67 us): a piece costs a prologue,
    // a partial row per query and its share of the merge launch.
    const int T = ps_tiles(s.nqb - 1, a.Sk, pcoff) + (a.causal && s.nqb > 1 ? ps_tiles(0, a.Sk, pcoff) : 0);
    long long n = slots / (pairs > 0 ? pairs : 1);
    n = n < ps_split_max_pieces() ? n : ps_split_max_pieces();
    n = n < T / kSplitMinTiles ? n : T / kSplitMinTiles;
    s.n = (int)n;
    s.nitems = pairs * s.n;
    s.bytes = (size_t)s.n * a.B * a.Hq * a.Sq * (size_t)(a.D + kPartPad) * sizeof(float);
    if (s.n < 2) return s;
    int ncut = 0;   // pairs that do get cut
    for (int near = 0; near < s.nwork; ++near) {
        const PSPair pr = ps_cuts(a.causal ? s.nqb - 1 - near : near, near, a.Sk, pcoff, s.n, ps_magic(s.n));
        int pieces = 0;
        for (int j = 0; j < s.n; ++j) pieces += pr.b[j + 1] > pr.b[j];
        ncut += pieces >= 2;
    }
    s.ok = 2 * ncut >= s.nwork;
    return s;
}

template <class T, int D, bool RAWOK>
int launch_ps_split(const FwdArgs& a, hipStream_t stream) {
    const PSSplitPlan s = ps_split_plan(a, device_cu_count(a.device));
    if (a.query_ws != nullptr) {
        *a.query_ws = s.bytes;
        return 0;
    }
    ScopedWorkspace ws(s.bytes, a.ws, a.ws_bytes, stream);
    if (ws.err != hipSuccess) return (int)ws.err;
    FwdPSParams p{};
    p.q = a.q; p.k = a.k; p.v = a.v; p.o = a.o; p.lse = a.lse;
    p.B = a.B; p.Hq = a.Hq; p.Hkv = a.Hkv; p.Sq = a.Sq; p.Sk = a.Sk;
    float c = a.scale * kLog2e;
    p.negq = c < 0.f;
    c = c < 0.f ? -c : c;
    if (c == 0.f) c = 1e-30f;
    p.c = c;
    p.nqb = s.nqb; p.pair = a.causal ? 1 : 0; p.nwork = s.nwork; p.coff = a.causal ? a.coff : 0;
    p.pcoff = a.causal ? a.coff : kEverything;
    p.nitems = (int)s.nitems;
    p.part = static_cast<float*>(ws.ptr);
    p.part_rows = a.B * a.Hq * a.Sq;
    p.npiece = s.n;
    p.magic = ps_magic(s.n);
    const size_t lds = ps_tile_lds<D>() + kMaxSlot * 20 + 16;
    if (a.causal)
        hipLaunchKernelGGL((fa_fwd_ps_kernel<T, D, true, RAWOK, false, false, true>), dim3((unsigned)s.nitems), dim3(512), lds, stream, p);
    else
        hipLaunchKernelGGL((fa_fwd_ps_kernel<T, D, false, RAWOK, false, false, true>), dim3((unsigned)s.nitems), dim3(512), lds, stream, p);
    int rc = (int)hipGetLastError();
    if (rc != 0) return rc;
    constexpr int RPW = 256 / (D / 4);
    const dim3 cgrid(kQBlock / RPW, (unsigned)s.nqb, (unsigned)(a.B * a.Hq));
    switch (s.n) {
        case 2: hipLaunchKernelGGL((fa_fwd_ps_combine<T, D, 2>), cgrid, dim3(256), 0, stream, p); break;
        case 3: hipLaunchKernelGGL((fa_fwd_ps_combine<T, D, 3>), cgrid, dim3(256), 0, stream, p); break;
        case 4: hipLaunchKernelGGL((fa_fwd_ps_combine<T, D, 4>), cgrid, dim3(256), 0, stream, p); break;
        case 5: hipLaunchKernelGGL((fa_fwd_ps_combine<T, D, 5>), cgrid, dim3(256), 0, stream, p); break;
        case 6: hipLaunchKernelGGL((fa_fwd_ps_combine<T, D, 6>), cgrid, dim3(256), 0, stream, p); break;
        case 7: hipLaunchKernelGGL((fa_fwd_ps_combine<T, D, 7>), cgrid, dim3(256), 0, stream, p); break;
        default: hipLaunchKernelGGL((fa_fwd_ps_combine<T, D, 8>), cgrid, dim3(256), 0, stream, p); break;
    }
    return (int)hipGetLastError();
}

template <class T, int D, bool RAWOK>
int set_attr_ps() {
    const int lds = ps_tile_lds<D>() + kMaxSlot * 20 + 16;
    int rc = (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&fa_fwd_ps_kernel<T, D, true, RAWOK>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    rc |= (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&fa_fwd_ps_kernel<T, D, false, RAWOK>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if constexpr (D >= 64) {
        rc |= (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&fa_fwd_ps_kernel<T, D, true, RAWOK, false, false, true>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        rc |= (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&fa_fwd_ps_kernel<T, D, false, RAWOK, false, false, true>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    }
    if constexpr (RAWOK && D >= 64) {
        rc |= (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&fa_fwd_ps_kernel<T, D, true, true, false, true>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        rc |= (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&fa_fwd_ps_kernel<T, D, false, true, false, true>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    }
    return rc;
}

static bool raw_softmax_enabled() {
    static const int v = [] {
        const char* e = getenv("AULE_HIP_FWD_SOFTMAX");
        return (e != nullptr && e[0] == 'c') ? 0 : 1;
    }();
    return v == 1;
}

}  // namespace

#ifdef AULE_DEBUG_HOOKS
// Debug: the bf16 D = 128 kernel with tagged s_memtime stamps of workgroup 0 (tools/timeline_ps.py).
int launch_fwd_ps_timeline(const FwdArgs& a, unsigned long long* dbg, hipStream_t stream) {
    if (a.dtype != kBF16 || a.D != 128) return -1;
    FwdPSParams p;
    p.q = a.q; p.k = a.k; p.v = a.v; p.o = a.o; p.lse = a.lse;
    p.B = a.B; p.Hq = a.Hq; p.Hkv = a.Hkv; p.Sq = a.Sq; p.Sk = a.Sk;
    p.c = a.scale * kLog2e; p.negq = 0;
    p.nqb = (a.Sq + kQBlock - 1) / kQBlock;
    p.pair = a.causal ? 1 : 0;
    p.nwork = p.pair ? (p.nqb + 1) / 2 : p.nqb;
    p.coff = a.causal ? a.coff : 0;
    p.nitems = p.nwork * a.B * a.Hq;
    p.dbg = dbg;
    const long long ncu = device_cu_count(a.device);
    const long long rounds = (p.nitems + ncu * kMaxItems - 1) / (ncu * kMaxItems);
    long long G = ncu * rounds;
    if (G > p.nitems) G = p.nitems;
    const dim3 grid((unsigned)G), block(512);
    const size_t lds = ps_tile_lds<128>() + kMaxSlot * 20 + 16;
    auto go = [&](auto kern) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        hipLaunchKernelGGL(kern, grid, block, lds, stream, p);
    };
    if (a.causal) go(&fa_fwd_ps_kernel<Bf16Traits, 128, true, true, true>);
    else go(&fa_fwd_ps_kernel<Bf16Traits, 128, false, true, true>);
    return (int)hipGetLastError();
}

#endif  // AULE_DEBUG_HOOKS

// Shapes the persistent tile stream takes (everything else: fa_fwd_pp_gfx950.hip).
bool fwd_ps_applicable(const FwdArgs& a) {
    if (a.dtype != kBF16 && a.dtype != kF16) return false;
    if (a.D != 32 && a.D != 64 && a.D != 128) return false;
    if (a.window > 0) return false;
    if (a.causal && a.coff < 0) return false;
    // every part needs >= 4 KV tiles (the staging cursors run at most three tiles ahead of the compute cursor and may
    // cross one seam, not two): the shortest part is the first Q block
    const long long first = a.causal ? ((long long)kQBlock + a.coff < a.Sk ? (long long)kQBlock + a.coff : a.Sk) : a.Sk;
    if (first <= 3 * kKVTile) return false;
    // row offsets are 32-bit in the part table, byte offsets inside one head 32-bit in the buffer descriptors
    if ((long long)a.B * a.Hq * a.Sq >= (1LL << 31) || (long long)a.B * a.Hkv * a.Sk >= (1LL << 31)) return false;
    if ((long long)a.Sq * a.D * 2 >= (1LL << 32) || (long long)a.Sk * a.D * 2 >= (1LL << 32)) return false;
    return true;
}

// Shapes whose Q rotation the stream kernel fuses (FwdArgs::rope_*): the fixed-reference instances of D = 64 / 128, tables
// readable as 16-byte rows through a 32-bit buffer descriptor.
bool fwd_ps_rope_fusable(const FwdArgs& a) {
    if (!fwd_ps_applicable(a) || !raw_softmax_enabled() || (a.D != 64 && a.D != 128)) return false;
    if (a.rope_cos == nullptr || a.rope_sin == nullptr || a.rope_pitch < a.D / 2 || (a.rope_pitch & 3) != 0) return false;
    if ((reinterpret_cast<uintptr_t>(a.rope_cos) | reinterpret_cast<uintptr_t>(a.rope_sin)) & 15) return false;
    if (a.rope_pos < 0 || (long long)a.Sq + a.rope_pos > a.rope_rows) return false;
    // rows of padding lanes (up to the end of the last 256-row block) index past the table: their offsets must not wrap
    const long long last = ((long long)(a.Sq + kQBlock - 1) / kQBlock * kQBlock + a.rope_pos) * a.rope_pitch * 4;
    return (long long)a.rope_rows * a.rope_pitch * 4 < (1LL << 32) && last < (1LL << 32);
}

// Small causal grids for the SPLIT instances (route 7).
bool fwd_ps_split_applicable(const FwdArgs& a) {
    if (ps_split_max_pieces() < 2 || (a.D != 64 && a.D != 128) || a.rope_cos != nullptr || !fwd_ps_applicable(a)) return false;
    if ((long long)a.Sk >= 65535LL * kKVTile) return false;                               // tile indices are 16-bit in the table
    if ((long long)a.Sq * (a.D + kPartPad) * 4 >= (1LL << 32)) return false;              // partial rows of a head: 32-bit offsets
    return ps_split_plan(a, device_cu_count(a.device)).ok;
}

int launch_fwd_ps_split(const FwdArgs& a, hipStream_t stream) {
    const bool raw = raw_softmax_enabled();
    if (a.dtype == kBF16) {
        if (a.D == 128) return raw ? launch_ps_split<Bf16Traits, 128, true>(a, stream) : launch_ps_split<Bf16Traits, 128, false>(a, stream);
        if (a.D == 64) return raw ? launch_ps_split<Bf16Traits, 64, true>(a, stream) : launch_ps_split<Bf16Traits, 64, false>(a, stream);
    } else if (a.dtype == kF16) {
        if (a.D == 128) return raw ? launch_ps_split<F16Traits, 128, true>(a, stream) : launch_ps_split<F16Traits, 128, false>(a, stream);
        if (a.D == 64) return raw ? launch_ps_split<F16Traits, 64, true>(a, stream) : launch_ps_split<F16Traits, 64, false>(a, stream);
    }
    return -1;
}

// Host view of the SPLIT plan for the CPU tests (aule_hip_debug_forward_split_plan): out = {n, nwork, then per pair
// ntf, ntn, b[0 .. kMaxPieces]}; returns the ints written, 0 when the shape does not take the path.
int fwd_ps_split_plan_dump(const FwdArgs& a, int* out, int cap) {
    if (!fwd_ps_split_applicable(a)) return 0;
    const PSSplitPlan s = ps_split_plan(a, device_cu_count(a.device));
    const int per = 2 + kMaxPieces + 1, need = 2 + s.nwork * per;
    if (out == nullptr || cap < need) return -need;
    out[0] = s.n; out[1] = s.nwork;
    for (int near = 0; near < s.nwork; ++near) {
        const PSPair pr = ps_cuts(a.causal ? s.nqb - 1 - near : near, near, a.Sk, a.causal ? a.coff : kEverything, s.n, ps_magic(s.n));
        int* o = out + 2 + near * per;
        o[0] = pr.ntf; o[1] = pr.ntn;
        for (int j = 0; j <= kMaxPieces; ++j) o[2 + j] = pr.b[j];
    }
    return need;
}

int launch_fwd_ps(const FwdArgs& a, hipStream_t stream) {
    if (a.dtype == kBF16) {
        if (raw_softmax_enabled()) {
            if (a.D == 128) return launch_ps<Bf16Traits, 128, true>(a, stream);
            if (a.D == 64) return launch_ps<Bf16Traits, 64, true>(a, stream);
            if (a.D == 32) return launch_ps<Bf16Traits, 32, true>(a, stream);
        } else {
            if (a.D == 128) return launch_ps<Bf16Traits, 128, false>(a, stream);
            if (a.D == 64) return launch_ps<Bf16Traits, 64, false>(a, stream);
            if (a.D == 32) return launch_ps<Bf16Traits, 32, false>(a, stream);
        }
    } else if (a.dtype == kF16) {
        if (raw_softmax_enabled()) {
            if (a.D == 128) return launch_ps<F16Traits, 128, true>(a, stream);
            if (a.D == 64) return launch_ps<F16Traits, 64, true>(a, stream);
            if (a.D == 32) return launch_ps<F16Traits, 32, true>(a, stream);
        } else {
            if (a.D == 128) return launch_ps<F16Traits, 128, false>(a, stream);
            if (a.D == 64) return launch_ps<F16Traits, 64, false>(a, stream);
            if (a.D == 32) return launch_ps<F16Traits, 32, false>(a, stream);
        }
    }
    return -1;
}

int configure_fwd_ps() {
    return set_attr_ps<Bf16Traits, 128, true>() | set_attr_ps<Bf16Traits, 64, true>() | set_attr_ps<Bf16Traits, 32, true>() |
           set_attr_ps<Bf16Traits, 128, false>() | set_attr_ps<Bf16Traits, 64, false>() | set_attr_ps<Bf16Traits, 32, false>() |
           set_attr_ps<F16Traits, 128, false>() | set_attr_ps<F16Traits, 64, false>() | set_attr_ps<F16Traits, 32, false>() |
           set_attr_ps<F16Traits, 128, true>() | set_attr_ps<F16Traits, 64, true>() | set_attr_ps<F16Traits, 32, true>();
}

}  // namespace aule_hip
