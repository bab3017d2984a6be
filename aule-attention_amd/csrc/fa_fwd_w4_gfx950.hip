// fa_fwd_w4_gfx950.hip -- FlashAttention-2 forward (16-bit I/O), ONE WAVE PER SIMD: workgroup = 4 waves x 64 query rows.
//
// Replaces python/aule/triton_flash_amd.py:97-240 (_flash_attn_fwd_amd; the tile shapes it autotunes over: :58-95) at the
// headline shapes.  The predecessors (fa_fwd_pp_gfx950.hip and the tile-stream kernel retired in round 4) put two 32-row waves on every SIMD: one
// ds_read_b128 per QK^T MFMA, two transpose reads per PV MFMA, two barriers per tile, 57 cycles per MFMA measured.  Here
//
//   * a wave owns the whole 512-register file of its SIMD: O^T (128 registers), the Q fragments (64) and ONE K tile (64)
//     live in accumulator registers, one V tile (64) in the top arch VGPRs, all named literally by the instruction
//     streams of fa_fwd_w4_asm.inc (generated: tools/gen_w4.py, register map in its docstring) -- the scores, the packed P and
//     the softmax temporaries too; hipcc gets v0-v51 for addresses, loop scalars and the epilogue (amdgpu_num_vgpr(52));
//   * a wave computes TWO 32-row blocks (A, B) against every K / V fragment it reads: 48 LDS reads per 64 MFMAs;
//   * the softmax runs in the gaps of the wave's own MFMAs: a tile step is phase 1 [S_{j+1} = K_{j+1} Q^T | softmax of
//     S_j[B] | V_j transpose reads | LDS-DMA requests] and phase 2 [O^T += V_j^T P_j^T | softmax of S_{j+1}[A] | K_{j+2}
//     reads], 32 MFMAs each, every filler placed by the generator; ONE barrier per tile;
//   * K / V tiles arrive by LDS-DMA into 3-deep rings, requested TWO tile steps before their first reader (K four tiles ahead
//     of the PV tile, V two): under load a request takes ~1.1 us to land, longer than one step -- the first build (2-deep rings,
//     vmcnt(0) at every step) waited ~500 of its 3100 cycles per step there (profiles/r3_w4_v1_timeline_*.txt).  A step waits
//     only for the PREVIOUS step's requests (s_waitcnt vmcnt(8)).  Images as in the predecessor (K rows with XOR-swizzled
//     16-byte chunks, V in [kv/4][d/16][4][16] sub-tiles);
//   * persistent grid, one workgroup per CU walking a list of 256-row Q blocks ("parts": the (n-1-i, i) causal pairs), the
//     next part's first tiles and its Q requested while this part finishes.
//
// Softmax: fixed reference: m_ref = row maximum of tile 0, P = exp2(S c - m_ref) for every tile, a range
// verdict on the row sums at the end of the part.  A part that fails it is run again after the stream with the EXACT
// row maximum as reference (one extra QK^T-only pass over its tiles), so the fast path is the only softmax code.
//
// Covers: bf16 / fp16, D = 128 / 64, causal (top-left or shifted by coff >= 0) and non-causal, scale != 0 (round 6: negative scales on negated Q
// fragments), any Sq, every part with at least 4 KV tiles, no window or (round 6: the WIN instances) a causal one of >= 128 keys; optional fused query
// rotation (half-split pairs) -- everything else stays on the predecessors.
#include <cstdlib>
#include <type_traits>

#include "fa_device.h"
#include "fa_kernels.h"
#include "fa_fwd_tile.h"
#include "fa_fwd_split.h"

namespace aule_hip {
namespace {

#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"   // (literal registers above the compiler's budget are "reserved": that is the point)
#ifndef W4_ASM_INC
#define W4_ASM_INC "fa_fwd_w4_asm.inc"
#endif
#include W4_ASM_INC   // (tools/w4_experiments.sh builds timing variants from differently generated streams)

struct FwdW4Params {
    const void* q;
    const void* k;
    const void* v;
    void* o;
    float* lse;
    int B, Hq, Hkv, Sq, Sk;
    float c;      // |scale| * log2(e) > 0
    int negq;     // scale < 0 (round 6): the Q fragments are negated in registers (W4Asm::negate_q) -- c s = |c| ((-q) k) exactly
    int nqb;      // 256-row Q blocks
    int nwork;    // work items per head: ceil(nqb/2) when pairing, else nqb
    int pair;     // item = Q blocks (nqb-1-i, i)
    int coff;     // causal position offset (query i sits at position i + coff)
    int nitems;   // nwork * B * Hq; workgroup g takes items g, g + gridDim.x, ...
    int rounds;   // > 0: "round order" of the causal part lists (below) with this many rounds, mper = heads per round
    int mper;
    // fused query rotation (half-split pairs, rope_gfx950.hip's arithmetic; K arrives rotated): tables [rrows][rpitch] fp32, query i
    // of a head reads row i + rpos; rcos == nullptr: none
    const float* rcos;
    const float* rsin;
    int rrows, rpitch, rpos;
    // small grids (fa_fwd_split.h): npiece > 1: an item is piece (item % npiece) of a pair of Q blocks; a part then covers a RANGE of
    // its block's key tiles and, unless the range is the whole block, leaves fp32 partial rows instead of O
    int npiece, pcoff;
    unsigned magic;
    float* part;       // [npiece][part_rows][D + kPartPad]
    int part_rows;     // B * Hq * Sq
    int window;        // sliding window (round 6, WIN instances only): query at position x sees keys x - window < k <= x; 0: none
    float sum_lo;      // lower bound of the range verdict on a row's sum of weights (w4_body: kSumLo)
    int wtail_min;     // WIN: whole tiles a wave needs between its left-edge tiles and its diagonal before its tail takes the embedded-request bodies
    int generic;       // AULE_HIP_W4_BODIES=generic: every step through the generic bodies (the embedded-request flow off: A/B, bit-identity test)
    unsigned long long* dbg;   // timeline build only: [4 waves][kW4TLMax] tagged s_memtime stamps of workgroup 0
};

constexpr int kW4TLMax = 2048;    // stamps per wave in the debug buffer
constexpr int kW4TLLds = 512;     // ... kept in LDS while the kernel runs (a global store per stamp would count in vmcnt and
                                  // make the step's counted wait cover one of its own requests)
constexpr int kW4MaxItems = 64;              // per workgroup (the host sizes the grid accordingly)
constexpr int kW4MaxSlot = 2 * kW4MaxItems;  // parts: two per item (the second one invalid for an unpaired block)

__device__ __forceinline__ int w4_rfl(int x) { return __builtin_amdgcn_readfirstlane(x); }

template <int N>
__device__ __forceinline__ float w4_acc_read() {
    float x = 0.f;
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("v_accvgpr_read_b32 %0, a%c1" : "=v"(x) : "n"(N));
#endif
    return x;
}

// O^T block (QB, d) -> registers -> scaled, rounded, written transposed into the slab row of this lane: lane (q, hi) owns columns
// 32 d + 8 g + 4 hi .. + 3 of row q in accumulator registers 4 g .. 4 g + 3 of block d
template <class T, int DB, int BASE, int I = 0>
__device__ __forceinline__ void w4_pack_block(char* dst, float inv) {
    if constexpr (I < DB * 4) {
        constexpr int d = I / 4, g4 = I % 4, N = BASE + d * 16 + 4 * g4;
        u32x2_t u;
        u[0] = T::pack2(w4_acc_read<N>() * inv, w4_acc_read<N + 1>() * inv);
        u[1] = T::pack2(w4_acc_read<N + 2>() * inv, w4_acc_read<N + 3>() * inv);
        *reinterpret_cast<u32x2_t*>(dst + (32 * d + 8 * g4) * 2) = u;
        w4_pack_block<T, DB, BASE, I + 1>(dst, inv);
    }
}

template <int D> constexpr int w4_lds_bytes() {   // 3 K + 3 V ring slots, one 32-row slab per wave, the part table
    return 6 * 64 * 2 * D + 4 * 32 * (2 * D + 16) + kW4MaxSlot * 20 + 16;
}

// WIN (round 6): causal sliding window.  A part is the range of its block's key tiles any of its rows sees (the part table's tile range, as
// for the key-range pieces); a WAVE starts at the first tile ITS 64 rows see (rounded down to an even position: the parity copies of the
// score / weight registers then line up with a part's own start): positions in front of it are idle ones, the position in front of its first
// tile runs the part prologue's body on that tile ("wave prologue": bare QK^T, references, P[A]); tiles that cross the window's left edge
// take the left-edge mask variants of the streams (SM 6: lo <= key; windows are at least two key tiles long, so no tile is cut on both sides).  The fixed reference of a row is the maximum of the wave's first
// tile under the CAUSAL mask only -- the keys in front of the window are real keys of the same head: a finite reference of the right size
// even for the rows whose window starts in the next tile -- and the exact-maximum stream takes the maximum under both bounds.
template <class T, int D, bool CAUSAL, bool TL, bool WIN = false>
__device__ __forceinline__ void w4_body(const FwdW4Params& p) {
    using A = W4Asm<T, D>;
    static_assert(!WIN || (CAUSAL && !TL && !A::PRE), "window instances: causal, no timeline build, no pre-scaled form");
    // Lower bound of the range verdict on a row's sum of weights.  Plain instances: a row's reference is the maximum over keys the row SEES (its
    // first tile), so its largest weight is >= 1 and only the upper bound can fail; 2^-100 catches empty sums.  WIN: the reference may come from
    // keys IN FRONT of the row's window and lie above everything the row sees -- every weight below 1.  bf16 weights keep their 8 bits down to
    // 2^-126; fp16 weights lose theirs below 2^-14 (found by tools/fuzz_parity.py window: a key 29 log2 units above the window's maximum,
    // outside it, left O = 0 for fp16 rows whose verdict passed): the sum must stay above 1/2 -- with up to 16 K visible keys the rounding of the
    // weights that fall below fp16's normal range (2^-25 each) then stays under 2^-10 of the result -- or the part goes to the exact-maximum stream.  On N(0, 1) logits
    // the sum is ~0.15 W: the bound is never near.
    // (the bound itself: FwdW4Params::sum_lo, read where a verdict is taken -- 1/2 for the fp16 window instances: a row with ONE visible key sums to exactly 1)
    using std::integral_constant;
    constexpr int RB = 2 * D, RBP = RB + 16, CPR = RB / 16, KS = D / 16, DB = D / 32;
    constexpr int KT = 64 * RB, VT = KT, NP = KT / 4096, NQ = 2 * KS;   // NQ: buffer loads of a wave's Q fragments
    constexpr int SLAB = 32 * RBP;   // one 32-row block of O, rows padded by 16 bytes
    constexpr int OFF_V = 3 * KT, OFF_SLAB = OFF_V + 3 * VT, TLDS = OFF_SLAB + 4 * SLAB;
    static_assert(A::NP == NP, "generator / kernel disagree on the tile geometry");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = w4_rfl(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    int4* const tab = reinterpret_cast<int4*>(smem + TLDS);                       // [kW4MaxSlot] {q row offset, kv row offset, qb | -1, -}
    int* const redo = reinterpret_cast<int*>(smem + TLDS + kW4MaxSlot * 16);      // [kW4MaxSlot] range verdicts, [kW4MaxSlot] = any
#if defined(__HIP_DEVICE_COMPILE__)
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
#else
    const unsigned lds0 = 0;
#endif

    // The tile loop keeps ~50 scalars live; the kernel arguments are NOT among them: everything outside the plain step reads them
    // through an opaque pointer (a fresh s_load from the kernarg segment instead of a register held for the whole kernel --
    // with them resident hipcc spilled scalars into vector lanes, then vectors to scratch, and every reload's vmcnt(0) drained
    // the LDS-DMA queue in the middle of the loop).
#if defined(__HIP_DEVICE_COMPILE__)
    typedef const __attribute__((address_space(4))) FwdW4Params* KernargPtr;   // (constant address space: scalar loads)
#else
    typedef const FwdW4Params* KernargPtr;
#endif
    auto P = [&]() __attribute__((always_inline)) -> KernargPtr {
        // (the parameter block is the kernel's only explicit argument: offset 0 of the kernarg segment.  Taking &p instead would
        // make clang copy the block to scratch.)
#if defined(__HIP_DEVICE_COMPILE__)
        const unsigned long long a = (unsigned long long)(uintptr_t)__builtin_amdgcn_kernarg_segment_ptr();
#else
        const unsigned long long a = 0;
#endif
        unsigned lo = (unsigned)a, hi = (unsigned)(a >> 32);
        asm volatile("" : "+s"(lo), "+s"(hi));   // (not hoistable; its results count as divergent, hence the readfirstlanes)
        lo = (unsigned)w4_rfl((int)lo);
        hi = (unsigned)w4_rfl((int)hi);
        return (KernargPtr)(uintptr_t)(((unsigned long long)hi << 32) | lo);
    };
    auto sum_lo = [&]() __attribute__((always_inline)) { return WIN ? P()->sum_lo : 0x1p-100f; };
    const int Sk = p.Sk, coff = p.coff;
    const int win = WIN ? p.window : 0;
    const float c = p.c;
    const bool rope = p.rcos != nullptr;
    // (negq is NOT held in a register: held, it cost the causal D = 128 instance 21 more scalar lane spills at its part boundaries -- the part
    // prologue re-reads it from the kernel arguments, once per part)
    auto negq = [&]() __attribute__((always_inline)) { return w4_rfl(P()->negq) != 0; };
    const bool embedded = p.generic == 0;
    // K / V base pointers stay in scalar registers for the whole kernel (round 4): a head's base is then a 64-bit add where the
    // cursors cross into the next part, not a load from the kernel-argument segment and its wait in the middle of a part's last steps
    const void* const kbase = p.k;
    const void* const vbase = p.v;
    // ... and so do the Q / O / LSE pointers and Sq: every part boundary builds three descriptors from them (hipcc parks them in
    // lanes of its spill register -- a v_readlane each -- which is still an order of magnitude cheaper than the kernel-argument loads)
    const void* const qbase = p.q;
    void* const obase = p.o;
    float* const lsebase = p.lse;
    const int Sq_k = p.Sq;   // AULE_HIP_W4_BODIES=generic turns the embedded-request bodies and the seam off
    int tl_n = 0;
    unsigned long long* const tl_lds = reinterpret_cast<unsigned long long*>(smem + TLDS + kW4MaxSlot * 20 + 16);   // TL only
    auto stamp = [&](int tag) __attribute__((always_inline)) {   // timeline build: (tag << 56) | shader clock
        if constexpr (TL) {
            if (blockIdx.x == 0 && tl_n < kW4TLLds) {
                const unsigned long long t = __builtin_amdgcn_s_memtime();
                if (lane == 0) tl_lds[wave * kW4TLLds + tl_n] = (t & 0x00ffffffffffffffull) | ((unsigned long long)tag << 56);
                ++tl_n;
            }
        }
    };
    if constexpr (TL) {
        for (int i = tid; i < 4 * kW4TLLds; i += 256) tl_lds[i] = 0;
    }

    // ---- part table.  Default: thread t describes part (t & 1) of this workgroup's item t >> 1 (item = the causal pair of Q
    // blocks (n-1-i, i) of one head: equal work per item).
    // Round order (p.rounds > 0; causal, the XCD's W = gridDim/8 workgroups a multiple of the n Q blocks of a head): the XCD's
    // heads are taken m = W/n at a time ("round"), workgroup w of the XCD gets Q block w/m of head w%m of the round in even
    // rounds and block n-1-w/m in odd ones -- two consecutive rounds are one causal pair per workgroup, so the lists stay
    // balanced, but only m heads' K/V (instead of 2m) are being walked by an XCD at any time, and the workgroups of an even
    // round start their parts together at tile 0: the K/V re-reads that missed the 4 MB L2 (profiles/r3_fwd_c2_*) mostly go.
    const int G = (int)gridDim.x;
    const int nit = (p.nitems - (int)blockIdx.x + G - 1) / G;
    const int nslot = (!WIN && p.rounds > 0) ? p.rounds : 2 * nit;
    if (tid < nslot) {
        int qb = -1, range = 0;   // .z = qb | (partial plane + 1) << 24 (0: O is final), .w = first tile | end tile << 16 (0: the whole block)
        WorkItem w;
        if (!WIN && p.npiece > 1) {   // (the window instances: no key-range pieces, no round order)
            // item = (pair, piece): slot 0 its range of the far block, slot 1 of the near block (non-causal: no pairing, far == near)
            w = decode_work((int)blockIdx.x + (tid >> 1) * G, p.B, p.Hq, p.Hkv, p.npiece * p.nwork, false);
            const int near = w.blk / p.npiece, piece = w.blk % p.npiece, far = p.pair ? p.nqb - 1 - near : near;
            const SplitPair pr = split_cuts(far, near, Sk, p.pcoff, p.npiece, p.magic);
            int t0, t1;
            split_range(pr, piece, tid & 1, t0, t1);
            if (t1 > t0) {
                const int whole = (tid & 1) ? pr.ntn : pr.ntf;
                qb = ((tid & 1) ? near : far) | ((t0 == 0 && t1 == whole) ? 0 : (piece + 1) << 24);
                range = t0 | (t1 << 16);
            }
        } else if (!WIN && p.rounds > 0) {
            const int W = G >> 3, wx = (int)blockIdx.x >> 3, pos = (tid & 1) ? W - 1 - wx : wx;
            const int g = p.Hq / p.Hkv;
            const int c = tid * p.mper + pos % p.mper;          // head of this XCD's list: kv unit c / g, head c % g of its group
            const int unit = ((int)blockIdx.x & 7) + 8 * (c / g);
            w.b = unit / p.Hkv;
            w.hk = unit % p.Hkv;
            w.h = w.hk * g + c % g;
            w.blk = pos / p.mper;
            qb = w.blk;
        } else {
            w = decode_work((int)blockIdx.x + (tid >> 1) * G, p.B, p.Hq, p.Hkv, p.nwork, false);
            if (p.pair) {
                const int far = p.nqb - 1 - w.blk;       // the larger block of the pair goes first
                if ((tid & 1) == 0) qb = far;
                else if (far != w.blk) qb = w.blk;
            } else if ((tid & 1) == 0) {
                qb = w.blk;
                if constexpr (WIN) {
                    // Which block of its head an item is, is free to choose per head (a rotation of the block indices of one head is a bijection whatever
                    // the grid): rotate by the head's position in the XCD's unit list.  Without it workgroup g gets block (g / 8) % nqb in EVERY item of
                    // its list (the items of a workgroup are a whole number of heads apart), and the workgroups that hold a head's first blocks -- whose
                    // windows are cut short by the start of the sequence -- idle while the others finish (W 1024 at S 8192: 4 of 32 block indices).
                    const int unit = w.b * p.Hkv + w.hk;
                    qb = (w.blk + (unit >> 3) * 13 + w.h * 7) % p.nqb;
                }
                if constexpr (WIN) {   // the key tiles the block's rows see: from its first row's first key to its last row's diagonal
                    // (a part has at least four tiles -- a ragged last block under a short window would have fewer: tiles in front of the window
                    // are masked like any other key outside it)
                    const int t1 = (max(1, min(Sk, qb * kQBlock + kQBlock + coff)) + kKVTile - 1) / kKVTile;
                    const int t0 = min(max(0, qb * kQBlock + coff - win + 1) / kKVTile, max(0, t1 - 4));
                    if (t0 > 0) range = t0 | (t1 << 16);
                }
            }
        }
        tab[tid] = int4{(w.b * p.Hq + w.h) * p.Sq, (w.b * p.Hkv + w.hk) * Sk, qb, range};
        redo[tid] = 0;
    }
    if (tid == 0) redo[kW4MaxSlot] = 0;
    __syncthreads();

    auto next_valid = [&](int slot) __attribute__((always_inline)) {
        do {
            ++slot;
        } while (slot < nslot && w4_rfl(tab[slot].z) < 0);
        return slot;
    };
    auto nt_of = [&](int qb) __attribute__((always_inline)) {
        const int kv_hi = CAUSAL ? max(1, min(Sk, qb * kQBlock + kQBlock + coff)) : Sk;
        return (kv_hi + kKVTile - 1) / kKVTile;
    };
    // (every piece of a descriptor goes through readfirstlane where it is built: hipcc moves uniform arithmetic to the vector
    // unit now and then -- the timeline build does -- and the requests want their descriptors in scalar registers)
    auto head_srd = [&](const void* base, int rowoff, int rows) __attribute__((always_inline)) {
        const unsigned long long a = (unsigned long long)(uintptr_t)base + (unsigned long long)(unsigned)rowoff * RB;
        const unsigned lo = (unsigned)w4_rfl((int)(unsigned)a), hi = (unsigned)w4_rfl((int)(unsigned)(a >> 32));
        return make_srd(reinterpret_cast<const void*>((uintptr_t)(((unsigned long long)hi << 32) | lo)), (unsigned)w4_rfl(rows * RB));
    };
    auto sq_of = [&]() __attribute__((always_inline)) { return Sq_k; };

    // ---- lane constants: LDS addresses of the operand reads, per-lane source offsets of the DMA pieces
    constexpr int SWSH = CPR == 16 ? 0 : 1;
    // K fragment (ks, h = 0) of ring slot 0: chunk (2 ks + hi) ^ swz(row) of row l31 = ka0 ^ 32 ks (the swizzle bits, the hi bit and
    // the row offset occupy disjoint bit ranges; slot offsets are multiples of the tile size); h = 1: + 32 rows.  ONE register:
    // the eight addresses of a tile are rebuilt where they are used (one v_xor each, what adding the slot offset cost anyway).
    const unsigned ka0 = lds0 + (unsigned)(l31 * RB + ((((l31 >> SWSH) & (CPR - 1)) ^ hi) * 16));
    auto kaddr = [&](unsigned base, int ks) __attribute__((always_inline)) { return base ^ (unsigned)(ks * 32); };
    const unsigned va = lds0 + OFF_V + (unsigned)(hi * (D / 16) * 128 + ((lane >> 4) & 1) * 128 + (lane & 15) * 8);
    // per-lane source offset of this wave's piece 0 of a K / V tile; piece i: + 4096 bytes in both maps (16 more rows of K; V: 32
    // sub-tiles further), which goes into the request's scalar offset
    unsigned kvo, vvo;
    {
        const int q = wave * 64 + lane, r = q / CPR, cs = q % CPR;
        kvo = (unsigned)(r * RB + (cs ^ ((r >> SWSH) & (CPR - 1))) * 16);
        const int bidx = q >> 3;
        vvo = (unsigned)(((bidx / (D / 16)) * 4 + ((q >> 1) & 3)) * RB + ((bidx % (D / 16)) * 2 + (q & 1)) * 16);
    }
    A::set_consts();
    const unsigned wave1k = (unsigned)wave * 1024u;
    A::set_lds_base((unsigned)w4_rfl((int)(lds0 + wave1k)));   // (literal scalar register of the embedded-request steps)
    const unsigned oob = (unsigned)Sk * RB;   // a scalar offset at which every lane of a request is out of range (LDS gets zeros)

    // The tile barrier of step j: the wave's LDS reads of step j - 1 have returned (so a request of step j may overwrite their
    // ring slots once every wave is here), and the tiles requested at step j - 2 have landed (everything but the NREQ pieces
    // step j - 1 asked for) -- every wave waits for its own pieces, the barrier makes them everybody's.  Generic and idle steps
    // run it first thing; the plain step carries it behind its second MFMA (fa_fwd_w4_asm.inc, phase-1 statement 0).
    auto step_begin = [&](auto nreq_tag) __attribute__((always_inline)) {
        constexpr int NREQ = decltype(nreq_tag)::value;
        static_assert(NREQ % 2 == 0 && NREQ <= 24, "vector-memory operations a step may leave in flight");
#define W4_STEP_BEGIN(N) asm volatile("s_waitcnt vmcnt(" #N ") lgkmcnt(0)\n\ts_barrier" ::: "memory")
        if constexpr (NREQ == 0) W4_STEP_BEGIN(0);
        else if constexpr (NREQ == 2) W4_STEP_BEGIN(2);
        else if constexpr (NREQ == 4) W4_STEP_BEGIN(4);
        else if constexpr (NREQ == 8) W4_STEP_BEGIN(8);
        else if constexpr (NREQ == 10) W4_STEP_BEGIN(10);
        else if constexpr (NREQ == 12) W4_STEP_BEGIN(12);
        else if constexpr (NREQ == 16) W4_STEP_BEGIN(16);
        else if constexpr (NREQ == 18) W4_STEP_BEGIN(18);
        else if constexpr (NREQ == 20) W4_STEP_BEGIN(20);
        else W4_STEP_BEGIN(24);
#undef W4_STEP_BEGIN
    };

    auto run_stream = [&](auto redo_tag) __attribute__((always_inline)) {
        constexpr bool REDO = decltype(redo_tag)::value != 0;
        int cs = next_valid(-1);
        if (cs >= nslot) return;

        // ---- part scalars
        int qoff, kvoff, qb, nt, nt3, na, jm, r0;   // nt3: nt rounded up to a multiple of 3 (idle steps pad the part: every part starts at ring phase 0)
        int f0 = 0, jl = 0;         // WIN: the wave's first position with arithmetic (even), the first tile no row of the wave sees cut by the window's left edge
        int tb = 0, pid = 0;        // first key tile of the part's range (kvoff already points at it), partial plane + 1 (0: O is final)
        unsigned nrec = oob;        // bytes of the head's K / V behind kvoff: the descriptors' bound (rows >= Sk read as zeros)
        int n_slot;
        bool pre = false;           // the next part's K_0..K_2, V_0, V_1 and Q ride along with this part's last steps
        // ring phase: stream position mod 3.  At step j (phase rp) V_j sits in slot rp, K_{j+2} in slot rp + 2, the step requests
        // K_{j+4} into slot rp + 1 and V_{j+2} into slot rp + 2 (all mod 3); positions run on through the parts.
        int rp = 0;
        bool q_asked = false;   // the next part's Q rows were requested inside this part's last tile
        int nprev = 0;   // pieces the previous step requested (the only vector-memory operations the next tile barrier leaves in flight)
        auto slot = [&](int d) __attribute__((always_inline)) { const int x = rp + d; return x >= 3 ? x - 3 : x; };
        // request cursors: what step j asks for (K tile j + 4, V tile j + 2); soff == oob: nothing
        // (the heads' base addresses are carried as two dwords and made into descriptors WHERE THEY ARE USED, through
        // readfirstlane: a descriptor held in a variable across the loop is what hipcc moves to vector registers when scalar
        // pressure rises -- the timeline build did -- and the requests want scalar operands)
        unsigned klo = 0, khi = 0, vlo = 0, vhi = 0;
        unsigned ksoff, vsoff;
        auto srd_of = [&](unsigned lo, unsigned hi) __attribute__((always_inline)) {
            const unsigned l = (unsigned)w4_rfl((int)lo), h = (unsigned)w4_rfl((int)hi);
            return make_srd(reinterpret_cast<const void*>((uintptr_t)(((unsigned long long)h << 32) | l)), (unsigned)w4_rfl((int)nrec));
        };
        // table entry of a part: K / V row of its first key tile, its Q block
        auto kv_row_of = [&](int sl) __attribute__((always_inline)) { return w4_rfl(tab[sl].y) + (w4_rfl(tab[sl].w) & 0xffff) * kKVTile; };
        auto qb_of = [&](int sl) __attribute__((always_inline)) { return w4_rfl(tab[sl].z) & 0xffffff; };
        auto head_lohi = [&](const void* base, int rowoff, unsigned& lo, unsigned& hi) __attribute__((always_inline)) {
            const unsigned long long a = (unsigned long long)(uintptr_t)base + (unsigned long long)(unsigned)rowoff * RB;
            lo = (unsigned)a;
            hi = (unsigned)(a >> 32);
        };
        // (K_3 of the next part is NOT prefetched: its ring slot is the one K_0 sits in until the next part's prologue has read it;
        // every part requests its own K_3 at its step 0)
        // (the heads' base addresses change twice per part -- at its start and when the cursor crosses into the next part -- and
        // are fetched from the kernel arguments only then: two scalar loads and their wait in every one of a part's last five
        // steps were ~300 cycles each with the matrix pipe idle)
        auto set_k = [&](int t) __attribute__((always_inline)) {   // tile t of this part, or tile t - nt3 < 3 of the next one
            ksoff = oob;
            if (t < nt) ksoff = (unsigned)t * KT;
            else if (pre && t >= nt3 && t - nt3 <= 3) {
                if (t == nt3) head_lohi(kbase, kv_row_of(n_slot), klo, khi);
                // (position 3 of the next part: its K_3 is not prefetched -- the slot holds K_0 until the next prologue has read
                // it -- but the embedded-request steps ask unconditionally, and an out-of-range request would put zeros there:
                // K_0 once more, the same bytes into the same slot)
                ksoff = t - nt3 == 3 ? 0u : (unsigned)(t - nt3) * KT;
            }
        };
        auto set_v = [&](int t) __attribute__((always_inline)) {   // ... or tile t - nt3 < 2 of the next one
            vsoff = oob;
            if (t < nt) vsoff = (unsigned)t * VT;
            else if (pre && t >= nt3 && t - nt3 < 2) {
                if (t == nt3) head_lohi(vbase, kv_row_of(n_slot), vlo, vhi);
                vsoff = (unsigned)(t - nt3) * VT;
            }
        };
        auto enter_part = [&](int sl) __attribute__((always_inline)) {
            const int4 e = tab[sl];
            qoff = w4_rfl(e.x);
            const int ez = w4_rfl(e.z), ew = w4_rfl(e.w);
            qb = ez & 0xffffff;
            pid = ez >> 24;
            tb = ew & 0xffff;                                   // (0 | 0: the whole block)
            nt = ew != 0 ? (ew >> 16) - tb : nt_of(qb);
            kvoff = w4_rfl(e.y) + tb * kKVTile;
            nrec = (unsigned)(Sk - tb * kKVTile) * RB;
            nt3 = (nt + 2) / 3 * 3;
            r0 = qb * kQBlock + wave * 64;
            const int vis = CAUSAL ? min(Sk, r0 + 64 + coff) : Sk;   // keys the wave's last row sees
            na = min(nt, max(1, (vis + kKVTile - 1) / kKVTile - tb));
            const int min_thr = CAUSAL ? min(r0 + coff, Sk - 1) : Sk - 1;   // keys EVERY row of the wave sees: 0 .. min_thr
            jm = max(0, ((min_thr + 1) >> 6) - tb);                         // first tile (of the range) that needs the mask
            // (opaque: without a mask these are the same for every part, and hipcc then evaluates every comparison of the step
            // logic once per kernel and keeps the ~25 results -- 64-bit lane masks -- alive across the stream: 48 scalar spills
            // and, with the spill lanes' own register, a vector spill to scratch in the non-causal D = 128 instance)
            asm volatile("" : "+s"(nt), "+s"(nt3), "+s"(na), "+s"(jm));
            if constexpr (WIN) {
                const int lo_first = max(0, r0 + coff - win + 1);        // first key the wave's FIRST row sees, ... its LAST row sees:
                const int lo_last = max(0, r0 + 63 + coff - win + 1);
                f0 = min(max(0, (lo_first >> 6) - tb), na - 1) & ~1;
                jl = max(0, ((lo_last + kKVTile - 1) >> 6) - tb);
                asm volatile("" : "+s"(f0), "+s"(jl));
            }
            n_slot = next_valid(sl);
            pre = !REDO && n_slot < nslot;
            q_asked = false;
            head_lohi(kbase, kvoff, klo, khi);
            head_lohi(vbase, kvoff, vlo, vhi);
            set_k(4);
            set_v(2);
            A::zero_sums();
        };
        // last visible key (minus 4 hi) of the lane's row in block QB, relative to tile j (recomputed where a mask is needed:
        // a handful of tiles per part)
        auto thr_of = [&](int qbsel, int j) __attribute__((always_inline)) {
            int lane_o = lane;
            asm volatile("" : "+v"(lane_o));
            const int row = r0 + 32 * qbsel + (lane_o & 31) + coff;
            return (CAUSAL ? min(row, Sk - 1) : Sk - 1) - (lane_o >> 5) * 4 - 64 * (j + tb);
        };
        // WIN: first visible key (minus 4 hi) of the lane's row in block QB, relative to tile j: key k of the tile is visible iff lo <= k (<= thr)
        // = thr - (window - 1): the window instances only take problems whose every query has its diagonal key inside Sk, so thr_of's clamp never
        // binds on a row that is stored (rows >= Sq: a window shifted left over real keys, results dropped) -- one lane value to keep, not two
        auto lo_from = [&](int thr) __attribute__((always_inline)) { return thr - (win - 1); };
        // SM code of tile j's softmax: 6 the window's left edge cuts it (key >= lo), 2 the diagonal does (key <= thr), 1 neither.  Never both: the
        // window instances take windows of at least two key tiles (fwd_w4_applicable), so the last row's first key lies at least a tile in front of
        // the first row's diagonal key: jl <= jm
        auto code_at = [&](int j) __attribute__((always_inline)) { return (WIN && j < jl) ? 6 : (j >= jm ? 2 : 1); };
        // after step j: the ring moves on, the cursors of step j + 1 (K tile j + 5, V tile j + 3)
        auto advance = [&](int j) __attribute__((always_inline)) {
            rp = slot(1);
            if (__builtin_expect(j + 5 < nt, 1)) {   // (one compare in the steady state)
                ksoff += KT;
                vsoff += VT;
                return;
            }
            set_k(j + 5);
            set_v(j + 3);
        };
        auto issue_q = [&](int q_off, int q_b) __attribute__((always_inline)) {   // rows >= Sq read as 0
            const __amdgpu_buffer_rsrc_t qrs = head_srd(qbase, q_off, sq_of());
            int lane_o = lane;
            asm volatile("" : "+v"(lane_o));
            const unsigned vo = (unsigned)((q_b * kQBlock + wave * 64 + (lane_o & 31)) * RB + (lane_o >> 5) * 16);
            A::load_q(qrs, vo, vo + 32 * RB);
        };
        // the cos / sin rows of the wave's 64 queries of Q block q_b -> the score / weight registers (dead between two parts);
        // rows beyond the table read as 0 (descriptor bounds): they belong to queries >= Sq, whose Q is 0 already
        auto issue_rope = [&](int q_b) __attribute__((always_inline)) {
            const KernargPtr pp = P();
            const unsigned tbytes = (unsigned)pp->rrows * (unsigned)pp->rpitch * 4u;
            const __amdgpu_buffer_rsrc_t crs = make_srd(pp->rcos, (unsigned)w4_rfl((int)tbytes)), srs = make_srd(pp->rsin, (unsigned)w4_rfl((int)tbytes));
            int lane_o = lane;
            asm volatile("" : "+v"(lane_o));
            const unsigned vo = (unsigned)(q_b * kQBlock + wave * 64 + (lane_o & 31) + pp->rpos) * (unsigned)(pp->rpitch * 4) + (unsigned)((lane_o >> 5) * 32);
            A::rope_request(crs, srs, vo, vo + 32u * (unsigned)(pp->rpitch * 4));
        };
        // - m_ref of block BLK from the scores of tile 0 (lane-local 32 values + the other half's)
        auto neg_ref = [&](auto blk_tag, bool masked, int thr) __attribute__((always_inline)) {
            constexpr int BLK = decltype(blk_tag)::value;
            float mx = masked ? A::template rowmax<BLK, 1>(thr) : A::template rowmax<BLK, 0>(thr);
            mx = fmaxf(mx, xhalf_fast(mx));
            return A::PRE ? -mx : -(mx * c);   // (pre form: the scores already carry c)
        };
        // (readfirstlane: hipcc sometimes moves the ring arithmetic to the vector unit; the requests want scalar registers)
        auto k_lds = [&]() __attribute__((always_inline)) { return (unsigned)w4_rfl((int)(lds0 + (unsigned)slot(1) * KT + wave1k)); };
        auto v_lds = [&]() __attribute__((always_inline)) { return (unsigned)w4_rfl((int)(lds0 + OFF_V + (unsigned)slot(2) * VT + wave1k)); };
        auto ring_lds = [&](unsigned base, int d) __attribute__((always_inline)) { return (unsigned)w4_rfl((int)(lds0 + base + (unsigned)slot(d) * KT + wave1k)); };

        // ---- tile step j (PAR = j & 1: the S[B] / P[A] register copies in use)
        // plain: nothing masked, not the first, not the last tile of the wave; the requests ride in the gaps (K pieces in phase 1,
        // V pieces in phase 2: an out-of-range request writes zeros into a ring slot that is free by construction)
        // (ring slots as immediates: SL = stream position mod 3.  Six bodies -- slot x parity -- so that between two MFMA
        // statements of a plain step hipcc has nothing to compute but the requests' scalar offsets: the first build with run-time
        // slots spent ~55 scalar instructions and half a dozen branches between two steps, with the matrix pipe idle)
        // (round 4: every scalar operand of the requests -- descriptors, tile offsets, LDS addresses -- sits in literal registers
        // above hipcc's budget (W4Asm::NS, set_cursors / set_lds_base); the streams advance the cursors themselves)
        const __amdgpu_buffer_rsrc_t nosrd = make_srd(nullptr, 0);   // (unused operand of the literal-register statements)
        auto plain = [&](auto sl_tag, auto par_tag) __attribute__((always_inline)) {
            constexpr int SL = decltype(sl_tag)::value, PAR = decltype(par_tag)::value;
            stamp(0x10 + PAR);
            A::template p1<0, PAR, 1, 1, 1, 2, SL>(c, va, 0, 0, nosrd, 0, kvo);
            A::template p1<1, PAR, 1, 1, 1, 2, SL>(c, va, 0, 0, nosrd, 0, kvo);
            A::template p1<2, PAR, 1, 1, 1, 2, SL>(c, va, 0, 0, nosrd, 0, kvo);
            A::template p1<3, PAR, 1, 1, 1, 2, SL>(c, va, 0, 0, nosrd, 0, kvo);
            stamp(0x18);
            constexpr int KSL = (SL + 2) % 3;
            A::template p2<0, PAR, 1, 1, 1, 2, KSL>(c, kaddr(ka0, 0), kaddr(ka0, KS / 4 - 1), 0, 0, nosrd, 0, vvo);
            A::template p2<1, PAR, 1, 1, 1, 2, KSL>(c, kaddr(ka0, KS / 4), kaddr(ka0, 2 * (KS / 4) - 1), 0, 0, nosrd, 0, vvo);
            A::template p2<2, PAR, 1, 1, 1, 2, KSL>(c, kaddr(ka0, 2 * (KS / 4)), kaddr(ka0, 3 * (KS / 4) - 1), 0, 0, nosrd, 0, vvo);
            A::template p2<3, PAR, 1, 1, 1, 2, KSL>(c, kaddr(ka0, 3 * (KS / 4)), kaddr(ka0, KS - 1), 0, 0, nosrd, 0, vvo);
            stamp(0x19);
        };
        // the step in front of the wave's last tile in the same form: S_{j+1}[A] is the masked tile
        auto prediag = [&](auto sl_tag, auto par_tag, int j) __attribute__((always_inline)) {
            constexpr int SL = decltype(sl_tag)::value, PAR = decltype(par_tag)::value;
            stamp(0x62 + PAR);
            A::template p1<0, PAR, 1, 1, 1, 2, SL>(c, va, 0, 0, nosrd, 0, kvo);
            A::template p1<1, PAR, 1, 1, 1, 2, SL>(c, va, 0, 0, nosrd, 0, kvo);
            A::template p1<2, PAR, 1, 1, 1, 2, SL>(c, va, 0, 0, nosrd, 0, kvo);
            A::template p1<3, PAR, 1, 1, 1, 2, SL>(c, va, 0, 0, nosrd, 0, kvo);
            stamp(0x18);
            constexpr int KSL = (SL + 2) % 3;
            const int tA = thr_of(0, j + 1);
            A::template p2<0, PAR, 1, 2, 1, 2, KSL>(c, kaddr(ka0, 0), kaddr(ka0, KS / 4 - 1), tA, 0, nosrd, 0, vvo);
            A::template p2<1, PAR, 1, 2, 1, 2, KSL>(c, kaddr(ka0, KS / 4), kaddr(ka0, 2 * (KS / 4) - 1), tA, 0, nosrd, 0, vvo);
            A::template p2<2, PAR, 1, 2, 1, 2, KSL>(c, kaddr(ka0, 2 * (KS / 4)), kaddr(ka0, 3 * (KS / 4) - 1), tA, 0, nosrd, 0, vvo);
            A::template p2<3, PAR, 1, 2, 1, 2, KSL>(c, kaddr(ka0, 3 * (KS / 4)), kaddr(ka0, KS - 1), tA, 0, nosrd, 0, vvo);
            stamp(0x19);
        };
        // the wave's last tile (no S_{j+1}): masked softmax of S_j[B] and O^T += V_j^T P_j^T in the overlapped form -- block A's half of
        // the PV rides with the softmax of block B (W4Asm::diag, tools/gen_w4.py:gen_diag); both requests, and the next part's Q rows
        // (the Q fragments are dead since the previous step's QK^T; rows >= Sq read as 0), ride along
        auto diag = [&](auto sl_tag, auto par_tag, int j) __attribute__((always_inline)) {
            constexpr int SL = decltype(sl_tag)::value, PAR = decltype(par_tag)::value;
            stamp(0x64 + PAR);
            const int tB = thr_of(1, j);
            A::template diag<0, PAR, SL, 0>(c, va, tB, kvo, 0, 0, nosrd);
            A::template diag<1, PAR, SL, 0>(c, va, tB, kvo, 0, 0, nosrd);
            A::template diag<2, PAR, SL, 0>(c, va, tB, kvo, 0, 0, nosrd);
            A::template diag<3, PAR, SL, 0>(c, va, tB, kvo, 0, 0, nosrd);
            stamp(0x18);
            if (pre) {
                const __amdgpu_buffer_rsrc_t qrs = head_srd(qbase, w4_rfl(tab[n_slot].x), sq_of());
                int lane_o = lane;
                asm volatile("" : "+v"(lane_o));
                const unsigned vo = (unsigned)((qb_of(n_slot) * kQBlock + wave * 64 + (lane_o & 31)) * RB + (lane_o >> 5) * 16);
                A::template diag<4, PAR, SL, 1>(c, va, tB, vvo, vo, vo + 32 * RB, qrs);
                A::template diag<5, PAR, SL, 1>(c, va, tB, vvo, vo, vo + 32 * RB, qrs);
            } else {
                A::template diag<4, PAR, SL, 0>(c, va, tB, vvo, 0, 0, nosrd);
                A::template diag<5, PAR, SL, 0>(c, va, tB, vvo, 0, 0, nosrd);
            }
            stamp(0x19);
        };
        // step 0 of a part (stream position 0, parity 0; O starts at 0; S_0 is the prologue's bare QK^T): its tile barrier has
        // nothing to wait for but the other waves' reads of K_0 and K_1 (the prologue's vmcnt(0) made the tiles visible)
        auto first = [&]() __attribute__((always_inline)) {
            stamp(0x60);
            A::template p1<0, 0, 1, 3, 1, 3, 0>(c, va, 0, 0, nosrd, 0, kvo);
            A::template p1<1, 0, 1, 3, 1, 2, 0>(c, va, 0, 0, nosrd, 0, kvo);
            A::template p1<2, 0, 1, 3, 1, 2, 0>(c, va, 0, 0, nosrd, 0, kvo);
            A::template p1<3, 0, 1, 3, 1, 2, 0>(c, va, 0, 0, nosrd, 0, kvo);
            stamp(0x18);
            A::template p2<0, 0, 2, 1, 1, 2, 2>(c, kaddr(ka0, 0), kaddr(ka0, KS / 4 - 1), 0, 0, nosrd, 0, vvo);
            A::template p2<1, 0, 2, 1, 1, 2, 2>(c, kaddr(ka0, KS / 4), kaddr(ka0, 2 * (KS / 4) - 1), 0, 0, nosrd, 0, vvo);
            A::template p2<2, 0, 2, 1, 1, 2, 2>(c, kaddr(ka0, 2 * (KS / 4)), kaddr(ka0, 3 * (KS / 4) - 1), 0, 0, nosrd, 0, vvo);
            A::template p2<3, 0, 2, 1, 1, 2, 2>(c, kaddr(ka0, 3 * (KS / 4)), kaddr(ka0, KS - 1), 0, 0, nosrd, 0, vvo);
            stamp(0x19);
        };
        // shadows -> literal registers (part start, and wherever the slow path moved them)
        auto sync_regs = [&]() __attribute__((always_inline)) {
            A::set_cursors((unsigned)w4_rfl((int)klo), (unsigned)w4_rfl((int)khi), (unsigned)w4_rfl((int)vlo), (unsigned)w4_rfl((int)vhi),
                           (unsigned)w4_rfl((int)nrec), (unsigned)w4_rfl((int)ksoff), (unsigned)w4_rfl((int)vsoff));
        };
        // after embedded-request step j: the streams moved both cursors one tile on; where the part's tiles end (its last five
        // steps) the cursors of step j + 1 come from set_k / set_v as before and overwrite them
        auto fix_cursors = [&](int j) __attribute__((always_inline)) {
            if (__builtin_expect(j + 5 >= nt, 0)) {
                set_k(j + 5);
                set_v(j + 3);
                if (pre && (j + 5 == nt3 || j + 3 == nt3)) sync_regs();   // the cursor crossed into the next part: its head's descriptor
                else A::set_offsets((unsigned)w4_rfl((int)ksoff), (unsigned)w4_rfl((int)vsoff));
            }
        };
        // a run of plain steps [j, jend), j = 1 (mod 6) at entry.  Every part starts at ring phase 0 (parts are padded to a
        // multiple of three positions), so step j of ANY part uses ring slot j mod 3 and parity j mod 2: the six bodies follow
        // each other in a fixed order and the loop between them is two compares and two branches -- a run-time choice among the
        // six came out of hipcc's structurizer as ~10 flag tests per step.
        auto plain_run = [&](int& j, int jend) __attribute__((always_inline)) {
            using I0 = integral_constant<int, 0>;
            using I1 = integral_constant<int, 1>;
            using I2 = integral_constant<int, 2>;
            for (;;) {
                if (j >= jend) break;
                plain(I1{}, I1{}); fix_cursors(j); ++j;
                if (j >= jend) break;
                plain(I2{}, I0{}); fix_cursors(j); ++j;
                if (j >= jend) break;
                plain(I0{}, I1{}); fix_cursors(j); ++j;
                if (j >= jend) break;
                plain(I1{}, I0{}); fix_cursors(j); ++j;
                if (j >= jend) break;
                plain(I2{}, I1{}); fix_cursors(j); ++j;
                if (j >= jend) break;
                plain(I0{}, I0{}); fix_cursors(j); ++j;
            }
        };
        // WIN: one plain step at ANY position (a window part's whole tiles start behind the wave's left-edge tiles, wherever those end)
        auto plain_one = [&](int j) __attribute__((always_inline)) {
            using I0 = integral_constant<int, 0>;
            using I1 = integral_constant<int, 1>;
            using I2 = integral_constant<int, 2>;
            switch (j % 6) {
                case 0: plain(I0{}, I0{}); break;
                case 1: plain(I1{}, I1{}); break;
                case 2: plain(I2{}, I0{}); break;
                case 3: plain(I0{}, I1{}); break;
                case 4: plain(I1{}, I0{}); break;
                default: plain(I2{}, I1{}); break;
            }
        };
        // the wave's last two tiles in the embedded-request form: step j (in front of the last tile) and step j + 1 (the last
        // tile); ring slot and parity follow from the position
        auto fast_tail = [&](int j) __attribute__((always_inline)) {
            using I0 = integral_constant<int, 0>;
            using I1 = integral_constant<int, 1>;
            using I2 = integral_constant<int, 2>;
            switch (j % 6) {
                case 0: prediag(I0{}, I0{}, j); fix_cursors(j); diag(I1{}, I1{}, j + 1); break;
                case 1: prediag(I1{}, I1{}, j); fix_cursors(j); diag(I2{}, I0{}, j + 1); break;
                case 2: prediag(I2{}, I0{}, j); fix_cursors(j); diag(I0{}, I1{}, j + 1); break;
                case 3: prediag(I0{}, I1{}, j); fix_cursors(j); diag(I1{}, I0{}, j + 1); break;
                case 4: prediag(I1{}, I0{}, j); fix_cursors(j); diag(I2{}, I1{}, j + 1); break;
                default: prediag(I2{}, I1{}, j); fix_cursors(j); diag(I0{}, I0{}, j + 1); break;
            }
            fix_cursors(j + 1);   // (the positions behind the last tile request in the same form: fast_pads)
        };
        // a position without arithmetic for this wave, embedded-request form: positions [j, nt3) behind the wave's last tile.  The
        // first one waits for everything but the last tile's requests (and the next part's Q rows it carried), a later one for
        // everything but its predecessor's; a padding position (j >= nt) has no readers and does not wait for tiles at all (what is
        // in flight there are the next part's first tiles: the next prologue's vmcnt(0) + barrier make them visible).
        auto fast_pads = [&](int j) __attribute__((always_inline)) {
            constexpr int WP = 2 * NP, WQ = 2 * NP + NQ;
            bool first_pad = true;
            for (; j < nt3; ++j) {
                stamp(0x08);
                const int sl = j % 3;
                const int wv = j >= nt ? -1 : (first_pad && pre ? WQ : WP);
#define W4_PAD(SLT)                                                                  \
    if (wv < 0) A::template pad<SLT, -1>(kvo, vvo);                                  \
    else if (wv == WP) A::template pad<SLT, WP>(kvo, vvo);                            \
    else A::template pad<SLT, WQ>(kvo, vvo);
                if (sl == 0) { W4_PAD(0) } else if (sl == 1) { W4_PAD(1) } else { W4_PAD(2) }
#undef W4_PAD
                fix_cursors(j);
                first_pad = false;
            }
            rp = 0;            // (parts are padded to a multiple of three positions)
            q_asked = pre;     // (the last tile carried the next part's Q rows)
        };
        // the requests of a step as separate statements (everywhere but the plain step); returns the pieces now in flight.  An
        // out-of-range request is skipped: at a part's last steps its ring slot may already hold a tile of the next part.
        auto requests = [&]() __attribute__((always_inline)) {
            int n = 0;
            if (ksoff != oob) { A::dma_tile(k_lds(), srd_of(klo, khi), (unsigned)w4_rfl((int)ksoff), kvo); n += NP; }
            if (vsoff != oob) { A::dma_tile(v_lds(), srd_of(vlo, vhi), (unsigned)w4_rfl((int)vsoff), vvo); n += NP; }
            return n;
        };
        auto begin_n = [&]() __attribute__((always_inline)) {
            if (nprev == 2 * NP) step_begin(integral_constant<int, 2 * NP>{});
            else if (nprev == NP) step_begin(integral_constant<int, NP>{});
            else if (nprev == 2 * NP + NQ) step_begin(integral_constant<int, 2 * NP + NQ>{});
            else if (nprev == NP + NQ) step_begin(integral_constant<int, NP + NQ>{});
            else if (nprev == NQ) step_begin(integral_constant<int, NQ>{});
            else step_begin(integral_constant<int, 0>{});
        };
        // generic step.  QK: S of tile j + 1 is computed (not the wave's last tile).  SMB / SMA: softmax of S_j[B] / S_{j+1}[A]: 1 plain,
        // 2 masked (SMA 0: none, with QK 0).  PV: 1, or 2 for tile 0 (O starts at 0).
        auto step = [&](auto par_tag, auto qk_tag, auto smb_tag, auto sma_tag, auto pv_tag, int j) __attribute__((always_inline)) {
            constexpr int PAR = decltype(par_tag)::value, QK = decltype(qk_tag)::value, SMB = decltype(smb_tag)::value;
            constexpr int SMA = decltype(sma_tag)::value, PV = decltype(pv_tag)::value;
            static_assert((QK == 0) == (SMA == 0), "the softmax of S_{j+1}[A] rides with its QK^T");
            const __amdgpu_buffer_rsrc_t ksrd = make_srd(nullptr, 0);   // (unused operand of the statements without requests)
            stamp(0x20 + PAR + 2 * QK + 4 * SMB);
            begin_n();
            if constexpr (PV == 2)   // tile 0: K_3 goes where K_0 was (this step's tile barrier freed the slot); it is older than this
                                     // step's own requests, so the NEXT tile barrier covers it.  (The embedded-request step 0 carries
                                     // the same four pieces in its first statement.)
                if (!WIN || j == 0)  // (WIN: a wave that starts late asked for its pieces of K_3 in its idle position 0)
                    A::dma_tile(ring_lds(0, 0), head_srd(kbase, kvoff, Sk - tb * kKVTile), 3u * KT, kvo);
            const int n = requests();
            const unsigned vap = va + (unsigned)rp * VT, kb = ka0 + (unsigned)slot(2) * KT;
            const int tB = SMB >= 2 ? thr_of(1, j) : 0;
            const int lB = SMB == 6 ? lo_from(tB) : 0;
            constexpr int SMB1 = (PV == 2 && SMB != 6) ? SMB + 2 : SMB;   // tile 0: S_0 is the prologue's bare QK^T (streams: SM 3 / 4; the window form has no pre variant)
            A::template p1<0, PAR, QK, SMB1, 1, 0>(c, vap, tB, 0, ksrd, 0, 0, lB);
            A::template p1<1, PAR, QK, SMB1, 1, 0>(c, vap, tB, 0, ksrd, 0, 0, lB);
            A::template p1<2, PAR, QK, SMB1, 1, 0>(c, vap, tB, 0, ksrd, 0, 0, lB);
            A::template p1<3, PAR, QK, SMB1, 1, 0>(c, vap, tB, 0, ksrd, 0, 0, lB);
            stamp(0x18);
            const int tA = SMA >= 2 ? thr_of(0, j + 1) : 0;
            const int lA = SMA == 6 ? lo_from(tA) : 0;
            A::template p2<0, PAR, PV, SMA, 1, 0>(c, kaddr(kb, 0), kaddr(kb, KS / 4 - 1), tA, 0, ksrd, 0, 0, lA);
            A::template p2<1, PAR, PV, SMA, 1, 0>(c, kaddr(kb, KS / 4), kaddr(kb, 2 * (KS / 4) - 1), tA, 0, ksrd, 0, 0, lA);
            A::template p2<2, PAR, PV, SMA, 1, 0>(c, kaddr(kb, 2 * (KS / 4)), kaddr(kb, 3 * (KS / 4) - 1), tA, 0, ksrd, 0, 0, lA);
            A::template p2<3, PAR, PV, SMA, 1, 0>(c, kaddr(kb, 3 * (KS / 4)), kaddr(kb, KS - 1), tA, 0, ksrd, 0, 0, lA);
            stamp(0x19);
            advance(j);
            nprev = n;
        };
        // ... picked at run time: the wave's last tile is always run masked (harmless where nothing is)
        auto step_rt = [&](auto par_tag, auto pv_tag, int j) __attribute__((always_inline)) {
            using I0 = integral_constant<int, 0>;
            using I1 = integral_constant<int, 1>;
            using I2 = integral_constant<int, 2>;
            using I6 = integral_constant<int, 6>;
            if (j + 1 >= na) step(par_tag, I0{}, I2{}, I0{}, pv_tag, j);
            else if (WIN && j < jl) {   // a left-edge tile: followed by another one, a whole one, or the diagonal tile
                if (j + 1 < jl) step(par_tag, I1{}, I6{}, I6{}, pv_tag, j);
                else if (j + 1 >= jm) step(par_tag, I1{}, I6{}, I2{}, pv_tag, j);
                else step(par_tag, I1{}, I6{}, I1{}, pv_tag, j);
            }
            else if (j >= jm) step(par_tag, I1{}, I2{}, I2{}, pv_tag, j);
            else if (j + 1 >= jm) step(par_tag, I1{}, I1{}, I2{}, pv_tag, j);
            else step(par_tag, I1{}, I1{}, I1{}, pv_tag, j);
        };
        auto idle = [&](int j, bool kread = false) __attribute__((always_inline)) {   // a tile this wave does not see, or a padding position of the part
            stamp(0x08);
            // (a padding position has no readers: its barrier only keeps requests from overtaking the last tile's LDS reads.  It
            // does NOT wait for tiles -- what is in flight there are the next head's first tiles, first touches with ~2x the usual
            // latency -- the next prologue's vmcnt(0) + barrier make them visible before anything reads them.)
            if (j >= nt) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            else begin_n();
            stamp(0x09);
            if constexpr (WIN) {   // position 0 of a wave that starts late: its pieces of K_3 (what step 0 of the other waves carries)
                if (j == 0) A::dma_tile(ring_lds(0, 0), head_srd(kbase, kvoff, Sk - tb * kKVTile), 3u * KT, kvo);
            }
            int n = requests();
            if constexpr (WIN) {
                // two positions in front of a late wave's first tile f0: its K fragments, from the slot every other wave reads them from in
                // this step's phase 2 (K_{j+2}; the next step's request overwrites it: done before the next tile barrier's lgkmcnt(0))
                if (kread) {
                    unsigned kap[KS];
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) kap[ks] = kaddr(ka0 + (unsigned)slot(2) * KT, ks);
                    A::kread_all(kap);
                }
            }
            stamp(0x0a);
            // the wave's Q registers are free (its last QK^T is behind it): the next part's Q rows now, not at the seam, where the
            // four waves' 64 row-strided loads (one 16-byte chunk per lane and row: ~64 cache lines per instruction) queue up
            // behind each other for ~3000 cycles -- the waves that see the whole block are then alone on that path.  (Behind the
            // step's requests, and counted: the next tile barrier does not wait for them.)
            if (pre && j == na && !q_asked) {
                issue_q(w4_rfl(tab[n_slot].x), qb_of(n_slot));
                n += NQ;
            }
            stamp(0x0b);
            advance(j);
            stamp(0x0c);
            nprev = n;
        };
        // part prologue = "step -1" (K_0, K_1 of the part in the ring, Q requested): S_0, the references, P_0[A]; leaves K_1 in
        // the fragment registers
        // SEAM (round 4): when the prologue follows a finished part of the same stream, its bare QK^T carries that part's pack (O^T /
        // l, rounded, transposed into the wave's slab: W4Asm::seam) in its MFMA gaps, and the slab rows go out between the halves --
        // the pack used to run in the epilogue with no MFMA around, at a lone wave's ~8 cycles per instruction.  seam_tag 1: invA / invB
        // = 1 / l of the finished part's blocks, flush(QB) = its slab -> global stores.
        auto prologue = [&](auto seam_tag, float invA, float invB, auto&& flush) __attribute__((always_inline)) {
            constexpr bool SEAM = decltype(seam_tag)::value != 0;
            stamp(0x30);
            const __amdgpu_buffer_rsrc_t ksrd = make_srd(nullptr, 0);   // (unused operand)
            unsigned kap[KS];
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) kap[ks] = kaddr(ka0 + (unsigned)rp * KT, ks);
            // every wave's pieces of this part's first tiles (requested by the previous part's last steps, which no longer wait
            // for them) and the Q fragments; the epilogue's stores ride along
            asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
            A::kread_all(kap);
            if constexpr (!REDO) {
                if (rope) A::rope_rotate();          // (the second stream rotated in its exact-maximum pass)
                if (negq()) A::negate_q();           // (a negative scale: see FwdW4Params::negq)
                A::prescale_q(c);                    // (pre form only)
            }
            if constexpr (SEAM) {
                int lane_o = lane;
                asm volatile("" : "+v"(lane_o));
                const unsigned dstv = lds0 + OFF_SLAB + (unsigned)wave * SLAB + (unsigned)((lane_o & 31) * RBP + 8 * (lane_o >> 5));
                stamp(0x34);
                A::template seam<0>(invA, dstv);
                A::template seam<1>(invA, dstv);
                stamp(0x35);
                flush(integral_constant<int, 0>{});
                stamp(0x36);
                A::template seam<2>(invB, dstv);
                A::template seam<3>(invB, dstv);
                stamp(0x37);
                flush(integral_constant<int, 1>{});
            } else {
                A::template p1<0, 1, 1, 0, 0, 0>(c, va, 0, 0, ksrd, 0, 0);
                A::template p1<1, 1, 1, 0, 0, 0>(c, va, 0, 0, ksrd, 0, 0);
                A::template p1<2, 1, 1, 0, 0, 0>(c, va, 0, 0, ksrd, 0, 0);
                A::template p1<3, 1, 1, 0, 0, 0>(c, va, 0, 0, ksrd, 0, 0);
            }
            stamp(0x31);
            if constexpr (WIN) {
                if (f0 > 0) {   // a wave that starts late: its own prologue runs on tile f0, in the position in front of it (wave_prologue)
                    nprev = 0;
                    return;
                }
            }
            const int tA = thr_of(0, 0);
            if constexpr (!REDO) {
                A::template set_ref<0>(neg_ref(integral_constant<int, 0>{}, jm == 0, tA));
                A::template set_ref<1>(neg_ref(integral_constant<int, 1>{}, jm == 0, thr_of(1, 0)));
            }
            const unsigned kb = ka0 + (unsigned)slot(1) * KT;   // K_1
            if (WIN && 0 < jl) {
                const int lA = lo_from(tA);
                A::template p2<0, 1, 0, 6, 1, 0>(c, kaddr(kb, 0), kaddr(kb, KS / 4 - 1), tA, 0, ksrd, 0, 0, lA);
                A::template p2<1, 1, 0, 6, 1, 0>(c, kaddr(kb, KS / 4), kaddr(kb, 2 * (KS / 4) - 1), tA, 0, ksrd, 0, 0, lA);
                A::template p2<2, 1, 0, 6, 1, 0>(c, kaddr(kb, 2 * (KS / 4)), kaddr(kb, 3 * (KS / 4) - 1), tA, 0, ksrd, 0, 0, lA);
                A::template p2<3, 1, 0, 6, 1, 0>(c, kaddr(kb, 3 * (KS / 4)), kaddr(kb, KS - 1), tA, 0, ksrd, 0, 0, lA);
            } else if (jm == 0) {
                A::template p2<0, 1, 0, 2, 1, 0>(c, kaddr(kb, 0), kaddr(kb, KS / 4 - 1), tA, 0, ksrd, 0, 0);
                A::template p2<1, 1, 0, 2, 1, 0>(c, kaddr(kb, KS / 4), kaddr(kb, 2 * (KS / 4) - 1), tA, 0, ksrd, 0, 0);
                A::template p2<2, 1, 0, 2, 1, 0>(c, kaddr(kb, 2 * (KS / 4)), kaddr(kb, 3 * (KS / 4) - 1), tA, 0, ksrd, 0, 0);
                A::template p2<3, 1, 0, 2, 1, 0>(c, kaddr(kb, 3 * (KS / 4)), kaddr(kb, KS - 1), tA, 0, ksrd, 0, 0);
            } else {
                A::template p2<0, 1, 0, 1, 1, 0>(c, kaddr(kb, 0), kaddr(kb, KS / 4 - 1), tA, 0, ksrd, 0, 0);
                A::template p2<1, 1, 0, 1, 1, 0>(c, kaddr(kb, KS / 4), kaddr(kb, 2 * (KS / 4) - 1), tA, 0, ksrd, 0, 0);
                A::template p2<2, 1, 0, 1, 1, 0>(c, kaddr(kb, 2 * (KS / 4)), kaddr(kb, 3 * (KS / 4) - 1), tA, 0, ksrd, 0, 0);
                A::template p2<3, 1, 0, 1, 1, 0>(c, kaddr(kb, 3 * (KS / 4)), kaddr(kb, KS - 1), tA, 0, ksrd, 0, 0);
            }
            stamp(0x33);
            nprev = 0;   // (the vmcnt(0) above left nothing in flight; step 0's tile barrier follows: every wave holds K_0 and K_1 then,
                         // and step 0 requests K_3 and K_4 into their slots)
        };

        // WIN: the prologue of a wave whose first tile is f0 = j + 1 > 0 (j odd: the statements of "step -1", parity 1): the position's tile
        // barrier and requests, then the bare QK^T of tile f0 (its K fragments were read two positions earlier: idle(j - 1, true)), the
        // references, P_{f0}[A] next to the reads of K_{f0 + 1} (= K_{j+2}: ring slot rp + 2, like any step's phase 2)
        auto wave_prologue = [&](int j) __attribute__((always_inline)) {
            const __amdgpu_buffer_rsrc_t ksrd = make_srd(nullptr, 0);   // (unused operand)
            stamp(0x38);
            begin_n();
            const int n = requests();
            A::template p1<0, 1, 1, 0, 0, 0>(c, va, 0, 0, ksrd, 0, 0);
            A::template p1<1, 1, 1, 0, 0, 0>(c, va, 0, 0, ksrd, 0, 0);
            A::template p1<2, 1, 1, 0, 0, 0>(c, va, 0, 0, ksrd, 0, 0);
            A::template p1<3, 1, 1, 0, 0, 0>(c, va, 0, 0, ksrd, 0, 0);
            const int tA = thr_of(0, j + 1), lA = lo_from(tA);
            if constexpr (!REDO) {   // (causal mask only: see the comment at the top of w4_body)
                A::template set_ref<0>(neg_ref(integral_constant<int, 0>{}, jm <= j + 1, tA));
                A::template set_ref<1>(neg_ref(integral_constant<int, 1>{}, jm <= j + 1, thr_of(1, j + 1)));
            }
            const unsigned kb = ka0 + (unsigned)slot(2) * KT;
#define W4_WP_P2(SMC)                                                                                                     \
    A::template p2<0, 1, 0, SMC, 1, 0>(c, kaddr(kb, 0), kaddr(kb, KS / 4 - 1), tA, 0, ksrd, 0, 0, lA);                     \
    A::template p2<1, 1, 0, SMC, 1, 0>(c, kaddr(kb, KS / 4), kaddr(kb, 2 * (KS / 4) - 1), tA, 0, ksrd, 0, 0, lA);          \
    A::template p2<2, 1, 0, SMC, 1, 0>(c, kaddr(kb, 2 * (KS / 4)), kaddr(kb, 3 * (KS / 4) - 1), tA, 0, ksrd, 0, 0, lA);    \
    A::template p2<3, 1, 0, SMC, 1, 0>(c, kaddr(kb, 3 * (KS / 4)), kaddr(kb, KS - 1), tA, 0, ksrd, 0, 0, lA);
            const int cd = code_at(j + 1);
            if (cd == 6) { W4_WP_P2(6) } else if (cd == 2) { W4_WP_P2(2) } else { W4_WP_P2(1) }
#undef W4_WP_P2
            stamp(0x39);
            advance(j);
            nprev = n;
        };

        // ---- epilogue of a part: O = O^T / l, rounded, transposed through the wave's LDS slab (block A, then block B), whole-row
        //      stores; LSE; range verdict of the fixed reference
        auto epilogue = [&]() __attribute__((always_inline)) {
            stamp(0x40);
            asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");   // the last PV MFMAs -> v_accvgpr_read
            if (rope && pre) issue_rope(qb_of(n_slot));   // (the next prologue's vmcnt(0) covers them)
            int lane_o = lane;
            asm volatile("" : "+v"(lane_o));
            const int l31o = lane_o & 31, hio = lane_o >> 5;
            char* const slab = smem + OFF_SLAB + wave * SLAB;
            float* const lsep = lsebase;
            const __amdgpu_buffer_rsrc_t lrs = make_srd(lsep + (size_t)(unsigned)qoff, lsep != nullptr ? (unsigned)sq_of() * 4u : 0u);
            const __amdgpu_buffer_rsrc_t ors = head_srd(obase, qoff, sq_of());   // rows >= Sq are dropped by the bounds check
            bool bad = false;
            // a part that covers only a range of its block's keys: un-normalised O^T rows (fp32, straight from the accumulator
            // file), the reference in log2 units and the row sum -> plane pid - 1 of the workspace; fa_fwd_combine merges the planes
            auto partial = [&](auto qb_tag) __attribute__((always_inline)) {
                constexpr int QB = decltype(qb_tag)::value;
                constexpr int PP = (D + kPartPad) * 4;   // bytes per partial row
                const KernargPtr pp = P();
                float* const base = pp->part + ((size_t)(unsigned)(pid - 1) * (size_t)(unsigned)pp->part_rows + (size_t)(unsigned)qoff) * (D + kPartPad);
                const __amdgpu_buffer_rsrc_t prs = make_srd(base, (unsigned)sq_of() * (unsigned)PP);   // rows >= Sq fall outside
                float lt, nm;
                A::template get_sums<QB>(lt, nm);
                lt += xhalf_fast(lt);
                const unsigned roff = (unsigned)(r0 + 32 * QB + l31o) * PP;
                A::template partial_store<QB>(prs, roff + 16u * (unsigned)hio);
                const u32x2_t ml = {__builtin_bit_cast(unsigned, -nm), __builtin_bit_cast(unsigned, lt)};
                __builtin_amdgcn_raw_buffer_store_b64(ml, prs, hio == 0 ? (int)(roff + D * 4) : 0x7ffffff0, 0, 0);
#ifndef W4_X_NOVERDICT
                if constexpr (!REDO) bad = bad || !((lt > sum_lo()) && (lt < (T::kDType == 2 ? 0x1p110f : 0x1p15f)));
#endif
            };
            auto half = [&](auto qb_tag) __attribute__((always_inline)) {
                constexpr int QB = decltype(qb_tag)::value;
                float lt, nm;
                A::template get_sums<QB>(lt, nm);
                lt += xhalf_fast(lt);
                const float inv = __builtin_amdgcn_rcpf(lt);
                w4_pack_block<T, DB, QB * DB * 16>(slab + l31o * RBP + 8 * hio, inv);
                const float lse = (fast_log2(lt) - nm) * kLn2;
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, lse), lrs, hio == 0 ? (r0 + 32 * QB + l31o) * 4 : 0x7ffffff0, 0, 0);
#ifndef W4_X_NOVERDICT   // (timing experiments with garbage arithmetic: no second stream)
                if constexpr (!REDO) bad = bad || !((lt > sum_lo()) && (lt < (T::kDType == 2 ? 0x1p110f : 0x1p15f)));
#endif
                // (the wave's own LDS accesses are ordered: no barrier between the slab's writes, reads and next writes)
#pragma unroll
                for (int i0 = 0; i0 < CPR / 2; i0 += 2) {
                    u32x4_t x[2];
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        const int idx = (i0 + i) * 64 + lane_o, row = idx / CPR, cc = idx % CPR;
                        x[i] = *reinterpret_cast<const u32x4_t*>(slab + row * RBP + cc * 16);
                    }
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        const int idx = (i0 + i) * 64 + lane_o, row = idx / CPR, cc = idx % CPR;
                        __builtin_amdgcn_raw_buffer_store_b128(x[i], ors, (r0 + 32 * QB + row) * RB + cc * 16, 0, 0);
                    }
                }
            };
            if (!WIN && pid != 0) {
                partial(integral_constant<int, 0>{});
                stamp(0x42);
                partial(integral_constant<int, 1>{});
            } else {
                half(integral_constant<int, 0>{});
                stamp(0x42);
                half(integral_constant<int, 1>{});
            }
            if constexpr (!REDO) {
                if (__builtin_amdgcn_ballot_w64(bad) != 0 && lane == 0) redo[cs] = redo[kW4MaxSlot] = 1;
            }
            stamp(0x41);
        };

        // the part of the epilogue that stays in front of the seam: row sums -> 1 / l, LSE, the range verdict (the pack itself rides with
        // the next prologue's QK^T); then the next part's scalars
        auto seam_scalars = [&](float& invA, float& invB) __attribute__((always_inline)) {
            stamp(0x40);
            if (rope) issue_rope(qb_of(n_slot));   // (the next prologue's vmcnt(0) covers them; consumed before the seam statements)
            int lane_o = lane;
            asm volatile("" : "+v"(lane_o));
            const int l31o = lane_o & 31, hio = lane_o >> 5;
            float* const lsep = lsebase;
            const __amdgpu_buffer_rsrc_t lrs = make_srd(lsep + (size_t)(unsigned)qoff, lsep != nullptr ? (unsigned)sq_of() * 4u : 0u);
            bool bad = false;
            auto one = [&](auto qb_tag, float& inv) __attribute__((always_inline)) {
                constexpr int QB = decltype(qb_tag)::value;
                float lt, nm;
                A::template get_sums<QB>(lt, nm);
                lt += xhalf_fast(lt);
                inv = __builtin_amdgcn_rcpf(lt);
                const float lse = (fast_log2(lt) - nm) * kLn2;
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, lse), lrs, hio == 0 ? (r0 + 32 * QB + l31o) * 4 : 0x7ffffff0, 0, 0);
#ifndef W4_X_NOVERDICT
                bad = bad || !((lt > sum_lo()) && (lt < (T::kDType == 2 ? 0x1p110f : 0x1p15f)));
#endif
            };
            one(integral_constant<int, 0>{}, invA);
            one(integral_constant<int, 1>{}, invB);
            if (__builtin_amdgcn_ballot_w64(bad) != 0 && lane == 0) redo[cs] = redo[kW4MaxSlot] = 1;
        };

        // REDO only: the exact row maxima of the part, one QK^T-only pass over its tiles through ring slot 0
        auto max_pass = [&]() __attribute__((always_inline)) {
            const __amdgpu_buffer_rsrc_t ksrd = make_srd(nullptr, 0);   // (unused operand)
            float mA = -INFINITY, mB = -INFINITY;   // in units of c
            issue_q(qoff, qb);
            if (rope) issue_rope(qb);
            for (int j = 0; j < nt; ++j) {
                __syncthreads();
                A::dma_tile(lds0 + wave1k, head_srd(kbase, kvoff, Sk - tb * kKVTile), (unsigned)j * KT, kvo);
                asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
                if (j < na) {
                    unsigned kap[KS];
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) kap[ks] = kaddr(ka0, ks);
                    A::kread_all(kap);
                    if (j == 0) {
                        if (rope) A::rope_rotate();
                        if (negq()) A::negate_q();
                        A::prescale_q(c);
                    }
                    A::template p1<0, 1, 1, 0, 0, 0>(c, va, 0, 0, ksrd, 0, 0);
                    A::template p1<1, 1, 1, 0, 0, 0>(c, va, 0, 0, ksrd, 0, 0);
                    A::template p1<2, 1, 1, 0, 0, 0>(c, va, 0, 0, ksrd, 0, 0);
                    A::template p1<3, 1, 1, 0, 0, 0>(c, va, 0, 0, ksrd, 0, 0);
                    if constexpr (WIN) {   // the exact maximum of what the row SEES: both bounds (a tile in front of the window: -inf)
                        const int ta = thr_of(0, j), tb1 = thr_of(1, j);
                        float xa = A::template rowmax<0, 2>(ta, lo_from(ta)), xb = A::template rowmax<1, 2>(tb1, lo_from(tb1));
                        xa = fmaxf(xa, xhalf_fast(xa));
                        xb = fmaxf(xb, xhalf_fast(xb));
                        mA = fmaxf(mA, xa * c);
                        mB = fmaxf(mB, xb * c);
                    } else {
                        mA = fmaxf(mA, -neg_ref(integral_constant<int, 0>{}, true, thr_of(0, j)));
                        mB = fmaxf(mB, -neg_ref(integral_constant<int, 1>{}, true, thr_of(1, j)));
                    }
                }
            }
            A::template set_ref<0>(-mA);
            A::template set_ref<1>(-mB);
            __syncthreads();
        };

        // ---- the stream
        bool cold = true;
        bool seam_done = false;   // the part's prologue already ran, fused with the previous part's pack
        enter_part(cs);
        for (;;) {
            if (REDO || cold) {   // nothing of this part is in flight: K_0, K_1, K_2, V_0, V_1
                if constexpr (REDO) max_pass();
                else {
                    issue_q(qoff, qb);
                    if (rope) issue_rope(qb);
                }
                const __amdgpu_buffer_rsrc_t k0 = head_srd(kbase, kvoff, Sk - tb * kKVTile), v0 = head_srd(vbase, kvoff, Sk - tb * kKVTile);
                A::dma_tile(ring_lds(0, 0), k0, 0u, kvo);
                A::dma_tile(ring_lds(0, 1), k0, (unsigned)KT, kvo);
                A::dma_tile(ring_lds(0, 2), k0, 2u * KT, kvo);
                A::dma_tile(ring_lds(OFF_V, 0), v0, 0u, vvo);
                A::dma_tile(ring_lds(OFF_V, 1), v0, (unsigned)VT, vvo);
                asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
                cold = false;
            }
            // Embedded-request flow (round 4): wherever the wave sees at least three tiles and none before its last one needs the
            // mask -- every part of a causal or ragged problem but a head's first block -- step 0, the plain steps, the step in
            // front of the last tile and the last tile all run bodies with static ring slots, literal scalar operands and the
            // requests in their MFMA gaps.  Everything else (and the exact-maximum stream) takes the generic bodies.
            // WIN: the wave's head -- idle positions, its own prologue, step 0, the tiles that cross the window's left edge -- takes the generic
            // bodies and everything behind it (whole tiles, the step in front of the diagonal tile, the diagonal tile, the pads) the
            // embedded-request ones, when the diagonal is the only masked tile there
            const bool fast = !WIN && !REDO && embedded && na >= 3 && jm >= na - 1;   // (WIN: step 0 is always a generic one -- one flow less for hipcc's scalar budget)
            int jgen = na;        // WIN: generic steps run up to here
            bool wtail = false;   // ... and the embedded-request bodies take over
            if constexpr (WIN) {
                const int js = max(max(f0 + 1, jl), 1);
                // (at least four whole tiles: with the two a W = 256 part has per wave, the change of flow costs more than it saves -- S 8192 W 256
                // 354 -> 367 us, W 1024 665 -> 627 us, gpurun sessions r6_s9 / r6_s10)
                wtail = !fast && !REDO && embedded && jm >= na - 1 && js + w4_rfl(P()->wtail_min) <= na - 2;
                if (wtail) jgen = js;
            }
            if (!seam_done) prologue(integral_constant<int, 0>{}, 0.f, 0.f, [](auto) {});
            seam_done = false;
            using I0 = integral_constant<int, 0>;
            using I1 = integral_constant<int, 1>;
            using I2 = integral_constant<int, 2>;
            int j = 1;
            if (fast) {
                sync_regs();
                first();
                fix_cursors(0);
                plain_run(j, na - 2);
                fast_tail(j);
                fast_pads(na);
                j = nt3;
            } else {
                if (WIN && f0 > 0) {   // the wave starts at tile f0 (even, >= 2): idle positions, its K fragments, its own prologue, its first tile
                    for (j = 0; j < f0 - 2; ++j) idle(j);
                    idle(f0 - 2, true);
                    wave_prologue(f0 - 1);
                    step_rt(I0{}, I2{}, f0);
                    j = f0 + 1;
                } else {
                    step_rt(I0{}, I2{}, 0);   // tile 0 (O starts at 0)
                }
                while (j < jgen) {
                    if (j & 1) step_rt(I1{}, I1{}, j);
                    else step_rt(I0{}, I1{}, j);
                    ++j;
                }
                if constexpr (WIN) {
                    if (wtail) {
                        sync_regs();   // (the generic steps moved the shadows: cursors of step j into the literal registers)
                        for (; j < na - 2; ++j) {
                            plain_one(j);
                            fix_cursors(j);
                        }
                        fast_tail(j);
                        fast_pads(na);
                        j = nt3;
                    }
                }
            }
            for (; j < nt3; ++j) idle(j);   // (tiles the wave does not see, and the padding of the part to a multiple of three positions)
            if (pre && na == nt && !q_asked) issue_q(w4_rfl(tab[n_slot].x), qb_of(n_slot));   // (waves with idle steps asked in their first one;
                                                                                                   // the embedded-request flow: in the last tile)
            if constexpr (!REDO) {
                if (pre && (WIN || pid == 0) && embedded) {
                    // the seam: this part's pack inside the next part's prologue (pre: there IS a next part of this stream; pid 0: the
                    // part's O is final -- a partial part stores its accumulators as they are)
                    float invA, invB;
                    seam_scalars(invA, invB);
                    const int oq = qoff, or0 = r0;
                    cs = n_slot;
                    enter_part(cs);
                    const __amdgpu_buffer_rsrc_t ors = head_srd(obase, oq, sq_of());   // rows >= Sq are dropped by the bounds check
                    prologue(integral_constant<int, 1>{}, invA, invB, [&](auto qb_tag) __attribute__((always_inline)) {
                        constexpr int QB = decltype(qb_tag)::value;
                        int lane_o = lane;
                        asm volatile("" : "+v"(lane_o));
                        // (the wave's own LDS accesses are ordered: no barrier between the slab's writes, reads and next writes)
                        const unsigned la = lds0 + OFF_SLAB + (unsigned)wave * SLAB + (unsigned)((lane_o / CPR) * RBP + (lane_o % CPR) * 16);
                        const unsigned vo = (unsigned)((or0 + 32 * QB + lane_o / CPR) * RB + (lane_o % CPR) * 16);
                        A::slab_out(ors, la, vo, vo + 4096u);
                    });
                    seam_done = true;
                    continue;
                }
            }
            epilogue();
            if (n_slot >= nslot) break;
            cs = n_slot;
            enter_part(cs);
            if constexpr (REDO) __syncthreads();
        }
        stamp(0x50);
    };

    run_stream(std::integral_constant<int, 0>{});
    __syncthreads();   // every verdict posted, every LDS tile buffer idle
    if (w4_rfl(redo[kW4MaxSlot]) != 0) {
        __syncthreads();
        if (tid < nslot && redo[tid] == 0) tab[tid].z = -1;   // second, sparse stream: only the flagged parts
        __syncthreads();
        run_stream(std::integral_constant<int, 1>{});
    }
    if constexpr (TL) {
        __syncthreads();
        if (blockIdx.x == 0)
            for (int i = tid; i < 4 * kW4TLLds; i += 256) p.dbg[(i / kW4TLLds) * kW4TLMax + (i % kW4TLLds)] = tl_lds[i];
    }
}

// The kernels: hipcc's VGPR budget is the generator's NV (an attribute wants a literal: one wrapper per head size).
#ifndef W4_NS
#define W4_NS 88   // the embedded-request steps own s[88:101] (tools/gen_w4.py NS); amdgpu_num_sgpr(NS + 8) leaves hipcc s0 .. s(NS-1): the attribute counts VCC, FLAT_SCRATCH, XNACK_MASK and rounds to the allocation granule
#endif
#ifndef W4_NV_D64
#define W4_NV_D64 84   // (52 for streams generated with W4_PRE=1: tools/w4_variants.sh)
#endif
template <class T, bool CAUSAL, bool TL>
__global__ void __launch_bounds__(256, 1) __attribute__((amdgpu_num_vgpr(52), amdgpu_num_sgpr(W4_NS + 8))) fa_fwd_w4_kernel_d128(const FwdW4Params p) {
    static_assert(W4Asm<T, 128>::NV == 52, "amdgpu_num_vgpr of the D = 128 kernel must be the generator's NV");
    static_assert(W4Asm<T, 128>::NS == W4_NS, "amdgpu_num_sgpr must be the generator's NS");
    w4_body<T, 128, CAUSAL, TL>(p);
}
template <class T, bool CAUSAL, bool TL>
__global__ void __launch_bounds__(256, 1) __attribute__((amdgpu_num_vgpr(W4_NV_D64), amdgpu_num_sgpr(W4_NS + 8))) fa_fwd_w4_kernel_d64(const FwdW4Params p) {
    static_assert(W4Asm<T, 64>::NS == W4_NS, "amdgpu_num_sgpr must be the generator's NS");
    static_assert(W4Asm<T, 64>::NV == W4_NV_D64, "amdgpu_num_vgpr of the D = 64 kernel must be the generator's NV");
    w4_body<T, 64, CAUSAL, TL>(p);
}
// the sliding-window instances (causal)
template <class T>
__global__ void __launch_bounds__(256, 1) __attribute__((amdgpu_num_vgpr(52), amdgpu_num_sgpr(W4_NS + 8))) fa_fwd_w4_kernel_d128_win(const FwdW4Params p) {
    w4_body<T, 128, true, false, true>(p);
}
template <class T>
__global__ void __launch_bounds__(256, 1) __attribute__((amdgpu_num_vgpr(W4_NV_D64), amdgpu_num_sgpr(W4_NS + 8))) fa_fwd_w4_kernel_d64_win(const FwdW4Params p) {
    w4_body<T, 64, true, false, true>(p);
}
template <class T, int D>
constexpr auto w4_kernel_win() {
    if constexpr (D == 128) return &fa_fwd_w4_kernel_d128_win<T>;
    else return &fa_fwd_w4_kernel_d64_win<T>;
}
template <class T, int D, bool CAUSAL, bool TL>
constexpr auto w4_kernel() {
    if constexpr (D == 128) return &fa_fwd_w4_kernel_d128<T, CAUSAL, TL>;
    else return &fa_fwd_w4_kernel_d64<T, CAUSAL, TL>;
}

#pragma clang diagnostic pop

// AULE_HIP_W4_BODIES=generic: the round-3 flow (generic bodies for step 0 and the masked steps, requests as separate statements)
static int w4_generic_bodies() {
    static const int v = [] {
        const char* e = std::getenv("AULE_HIP_W4_BODIES");
        return (e != nullptr && e[0] == 'g') ? 1 : 0;
    }();
    return v;
}

template <class T, int D, bool TL = false>
int launch_w4(const FwdArgs& a, hipStream_t stream, unsigned long long* dbg = nullptr) {
    FwdW4Params p;
    p.q = a.q; p.k = a.k; p.v = a.v; p.o = a.o; p.lse = a.lse;
    p.B = a.B; p.Hq = a.Hq; p.Hkv = a.Hkv; p.Sq = a.Sq; p.Sk = a.Sk;
    p.c = (a.scale < 0.f ? -a.scale : a.scale) * kLog2e;
    p.negq = a.scale < 0.f ? 1 : 0;
    p.nqb = (a.Sq + kQBlock - 1) / kQBlock;
    p.pair = a.causal ? 1 : 0;
    p.nwork = p.pair ? (p.nqb + 1) / 2 : p.nqb;
    p.coff = a.causal ? a.coff : 0;
    p.nitems = p.nwork * a.B * a.Hq;
    p.dbg = dbg;
    p.npiece = 1; p.pcoff = 0; p.magic = 0; p.part = nullptr; p.part_rows = 0;
    p.generic = w4_generic_bodies();
    p.window = a.window > 0 ? a.window : 0;
    {   // AULE_HIP_W4_WTAIL=<n> (A/B; default 4: profiles/r6_window_shapes.txt)
        static const int wt = [] {
            const char* e = std::getenv("AULE_HIP_W4_WTAIL");
            return (e != nullptr && e[0] >= '0' && e[0] <= '9') ? std::atoi(e) : 4;
        }();
        p.wtail_min = wt;
        // (AULE_HIP_W4_SUMLO=<x>: A/B of the verdict's lower bound)
        static const float sl = [] {
            const char* e = std::getenv("AULE_HIP_W4_SUMLO");
            return e != nullptr ? (float)std::atof(e) : -1.0f;
        }();
        p.sum_lo = sl >= 0.f ? sl : ((p.window > 0 && a.dtype != kBF16) ? 0.5f : 0x1p-100f);
    }
    p.rcos = a.rope_cos; p.rsin = a.rope_sin;
    p.rrows = a.rope_rows; p.rpitch = a.rope_pitch; p.rpos = a.rope_pos;
    // one workgroup per CU; more only when a workgroup's list would not fit its part table
    const long long ncu = device_cu_count(a.device);
    // Half-empty causal grids (round 5; the reference harness's B 1 H 32 S 2048: 128 pairs on 256 CUs): when every Q block can have
    // a CU of its own, the blocks are NOT paired -- the launch then lasts as long as its largest block (32 tiles + one part's seam
    // instead of a pair's 36 tiles + two), on a chip that is ~56 % busy on average.  AULE_HIP_W4_UNPAIR=0 keeps the pairs (A/B).
    {
        static const int unpair = [] {
            const char* e = std::getenv("AULE_HIP_W4_UNPAIR");
            return (e != nullptr && e[0] == '0') ? 0 : 1;
        }();
        // (a sliding window: every block's part has about W / 64 + 4 tiles -- nothing to balance by pairing)
        if (a.causal && (p.window > 0 || (unpair && p.nqb >= 2 && (long long)p.nqb * a.B * a.Hq <= ncu))) {
            p.pair = 0;
            p.nwork = p.nqb;
            p.nitems = p.nwork * a.B * a.Hq;
        }
    }
    const long long rounds = (p.nitems + ncu * kW4MaxItems - 1) / (ncu * kW4MaxItems);
    long long G = ncu * rounds;
    if (G > p.nitems) G = p.nitems;
    p.rounds = 0;
    p.mper = 1;
    {   // round order (w4_body): needs whole rounds -- W a multiple of the heads' Q blocks -- and an even number of them
        static const char* const e = std::getenv("AULE_HIP_W4_ORDER");   // "pairs": the item order everywhere (A/B)
        const int units = a.B * a.Hkv, W = (int)(G / 8), g = a.Hq / a.Hkv;
        if (a.causal && p.window == 0 && !(e != nullptr && e[0] == 'p') && G == ncu && (G & 7) == 0 && (units & 7) == 0 && p.nqb <= W && W % p.nqb == 0) {
            const int m = W / p.nqb, hx = units / 8 * g;
            if (hx % (2 * m) == 0 && hx / m <= kW4MaxSlot) {
                p.rounds = hx / m;
                p.mper = m;
            }
        }
    }
    const dim3 grid((unsigned)G), block(256);
    const size_t lds = w4_lds_bytes<D>() + (TL ? 4 * kW4TLLds * 8 : 0);
    if (p.window > 0)
        hipLaunchKernelGGL((w4_kernel_win<T, D>()), grid, block, lds, stream, p);
    else if (a.causal)
        hipLaunchKernelGGL((w4_kernel<T, D, true, TL>()), grid, block, lds, stream, p);
    else
        hipLaunchKernelGGL((w4_kernel<T, D, false, TL>()), grid, block, lds, stream, p);
    return (int)hipGetLastError();
}

// Small grids: every pair of causal Q blocks (every non-causal block) as n work items of 1/n of its key tiles, partial rows, one
// merge launch (fa_fwd_split.h; route 7).  One workgroup per item.
template <class T, int D>
int launch_w4_split(const FwdArgs& a, hipStream_t stream) {
    const SplitPlan s = split_plan(a, device_cu_count(a.device));
    if (a.query_ws != nullptr) {
        *a.query_ws = s.bytes;
        return 0;
    }
    ScopedWorkspace ws(s.bytes, a.ws, a.ws_bytes, stream);
    if (ws.err != hipSuccess) return (int)ws.err;
    FwdW4Params p{};
    p.q = a.q; p.k = a.k; p.v = a.v; p.o = a.o; p.lse = a.lse;
    p.B = a.B; p.Hq = a.Hq; p.Hkv = a.Hkv; p.Sq = a.Sq; p.Sk = a.Sk;
    p.c = (a.scale < 0.f ? -a.scale : a.scale) * kLog2e;
    p.negq = a.scale < 0.f ? 1 : 0;
    p.nqb = s.nqb; p.pair = a.causal ? 1 : 0; p.nwork = s.nwork; p.coff = a.causal ? a.coff : 0;
    p.nitems = (int)s.nitems;
    p.rounds = 0; p.mper = 1;
    p.npiece = s.n; p.pcoff = a.causal ? a.coff : kEverything; p.magic = split_magic(s.n);
    p.part = static_cast<float*>(ws.ptr);
    p.part_rows = a.B * a.Hq * a.Sq;
    p.generic = w4_generic_bodies();
    p.window = 0; p.wtail_min = 0; p.sum_lo = 0x1p-100f;
    const size_t lds = w4_lds_bytes<D>();
    if (a.causal)
        hipLaunchKernelGGL((w4_kernel<T, D, true, false>()), dim3((unsigned)s.nitems), dim3(256), lds, stream, p);
    else
        hipLaunchKernelGGL((w4_kernel<T, D, false, false>()), dim3((unsigned)s.nitems), dim3(256), lds, stream, p);
    int rc = (int)hipGetLastError();
    if (rc != 0) return rc;
    CombineParams c{};
    c.o = a.o; c.lse = a.lse; c.part = p.part; c.part_rows = p.part_rows;
    c.Sq = a.Sq; c.Sk = a.Sk; c.nqb = s.nqb; c.pair = p.pair; c.pcoff = p.pcoff; c.npiece = s.n; c.magic = p.magic;
    return launch_combine<T, D>(c, a.B * a.Hq, stream);
}

template <class T, int D>
int set_attr_w4() {
    const int lds = w4_lds_bytes<D>();
    int rc = (int)hipFuncSetAttribute(reinterpret_cast<const void*>(w4_kernel<T, D, true, false>()),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    rc |= (int)hipFuncSetAttribute(reinterpret_cast<const void*>(w4_kernel<T, D, false, false>()),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    rc |= (int)hipFuncSetAttribute(reinterpret_cast<const void*>(w4_kernel_win<T, D>()), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    return rc;
}

}  // namespace

// Shapes the one-wave-per-SIMD forward takes (everything else: fa_fwd_pp_gfx950.hip).
bool fwd_w4_applicable(const FwdArgs& a) {
    if (a.dtype != kBF16 && a.dtype != kF16) return false;
    if (a.D != 128 && a.D != 64) return false;
    if (a.window > 0) {
        // sliding window (round 6, WIN instances): causal, every query with its own diagonal key inside Sk (no row without a visible key: the
        // references are taken from real scores), tile indices that fit the part table's 16 bits; AULE_HIP_W4_WINDOW=0: the ping-pong kernel (A/B)
        static const int on = [] {
            const char* e = std::getenv("AULE_HIP_W4_WINDOW");
            return (e != nullptr && e[0] == '0') ? 0 : 1;
        }();
        // (windows shorter than two key tiles stay where they were: a wave would see two or three tiles of a five-tile part, and a tile could be
        // cut by the window's left edge AND the diagonal -- the streams carry no mask variant with both bounds)
        if (!on || !a.causal || a.window < 2 * kKVTile || a.coff < 0 || (long long)a.Sq + a.coff > a.Sk || (long long)a.Sk >= 65535LL * kKVTile) return false;
    }
    if (a.rope_cos != nullptr) {   // fused query rotation: table geometry the 32-bit row offsets of the requests can address
        if (a.rope_sin == nullptr || a.rope_pitch < a.D / 2 || (a.rope_pitch & 3) != 0) return false;
        if ((reinterpret_cast<uintptr_t>(a.rope_cos) | reinterpret_cast<uintptr_t>(a.rope_sin)) & 15) return false;
        if (a.rope_pos < 0 || (long long)a.rope_rows < (long long)a.Sq + a.rope_pos) return false;
        // (rows of padding lanes -- up to the end of the last 256-row block -- index past the table: their offsets must not wrap)
        const long long last = ((long long)(a.Sq + kQBlock - 1) / kQBlock * kQBlock + a.rope_pos) * a.rope_pitch * 4;
        if ((long long)a.rope_rows * a.rope_pitch * 4 >= (1LL << 32) || last >= (1LL << 32)) return false;
    }
    {   // (round 6: negative scales run here too, on negated Q fragments; scale = 0 -- uniform attention -- stays on the ping-pong kernel)
        const float as = a.scale < 0.f ? -a.scale : a.scale;
        if (!(as > 0.f) || !(as < 3.0e38f)) return false;
    }
    if (a.causal && a.coff < 0) return false;
    // every part needs >= 4 KV tiles (its prologue consumes tiles 0 and 1 and requests tile 2 before the first plain step):
    // the shortest part is the first Q block
    const long long first = a.causal ? ((long long)kQBlock + a.coff < a.Sk ? (long long)kQBlock + a.coff : a.Sk) : a.Sk;
    if (first <= 3 * kKVTile) return false;
    // row offsets are 32-bit in the part table, byte offsets inside one head 32-bit in the buffer descriptors
    if ((long long)a.B * a.Hq * a.Sq >= (1LL << 31) || (long long)a.B * a.Hkv * a.Sk >= (1LL << 31)) return false;
    if ((long long)a.Sq * a.D * 2 >= (1LL << 31) || (long long)a.Sk * a.D * 2 >= (1LL << 31)) return false;
    return true;
}

// Small grids the forward cuts along the keys (route 7).
bool fwd_w4_split_applicable(const FwdArgs& a) {
    if (split_max_pieces() < 2 || a.rope_cos != nullptr || a.window > 0 || !fwd_w4_applicable(a)) return false;
    if ((long long)a.Sk >= 65535LL * kKVTile) return false;                               // tile indices are 16-bit in the part table
    if ((long long)a.Sq * (a.D + kPartPad) * 4 >= (1LL << 32)) return false;              // partial rows of a head: 32-bit offsets
    return split_plan(a, device_cu_count(a.device)).ok;
}

int launch_fwd_w4_split(const FwdArgs& a, hipStream_t stream) {
    if (a.dtype == kBF16 && a.D == 128) return launch_w4_split<Bf16Traits, 128>(a, stream);
    if (a.dtype == kF16 && a.D == 128) return launch_w4_split<F16Traits, 128>(a, stream);
    if (a.dtype == kBF16 && a.D == 64) return launch_w4_split<Bf16Traits, 64>(a, stream);
    if (a.dtype == kF16 && a.D == 64) return launch_w4_split<F16Traits, 64>(a, stream);
    return -1;
}

// Host view of the split plan for the CPU tests (aule_hip_debug_forward_split_plan): out = {n, nwork, then per pair ntf, ntn,
// b[0 .. kMaxPieces]}; returns the ints written, 0 when the shape does not take the path.
int fwd_split_plan_dump(const FwdArgs& a, int* out, int cap) {
    if (!fwd_w4_split_applicable(a)) return 0;
    const SplitPlan s = split_plan(a, device_cu_count(a.device));
    const int per = 2 + kMaxPieces + 1, need = 2 + s.nwork * per;
    if (out == nullptr || cap < need) return -need;
    out[0] = s.n; out[1] = s.nwork;
    for (int near = 0; near < s.nwork; ++near) {
        const SplitPair pr = split_cuts(a.causal ? s.nqb - 1 - near : near, near, a.Sk, a.causal ? a.coff : kEverything, s.n, split_magic(s.n));
        int* o = out + 2 + near * per;
        o[0] = pr.ntf; o[1] = pr.ntn;
        for (int j = 0; j <= kMaxPieces; ++j) o[2 + j] = pr.b[j];
    }
    return need;
}

int launch_fwd_w4(const FwdArgs& a, hipStream_t stream) {
    if (a.dtype == kBF16 && a.D == 128) return launch_w4<Bf16Traits, 128>(a, stream);
    if (a.dtype == kF16 && a.D == 128) return launch_w4<F16Traits, 128>(a, stream);
    if (a.dtype == kBF16 && a.D == 64) return launch_w4<Bf16Traits, 64>(a, stream);
    if (a.dtype == kF16 && a.D == 64) return launch_w4<F16Traits, 64>(a, stream);
    return -1;
}

#ifdef AULE_DEBUG_HOOKS
// Debug: the bf16 kernel (D = 128; D = 64 when built with -DW4_TL_D64) with tagged s_memtime stamps of workgroup 0
// (tools/timeline_w4.py).
#ifdef W4_TL_D64
constexpr int kW4TLD = 64;
#else
constexpr int kW4TLD = 128;
#endif
int launch_fwd_w4_timeline(const FwdArgs& a, unsigned long long* dbg, hipStream_t stream) {
    if (a.dtype != kBF16 || a.D != kW4TLD || !fwd_w4_applicable(a)) return -1;
    const int lds = w4_lds_bytes<kW4TLD>() + 4 * kW4TLLds * 8;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(w4_kernel<Bf16Traits, kW4TLD, true, true>()), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(w4_kernel<Bf16Traits, kW4TLD, false, true>()), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    return launch_w4<Bf16Traits, kW4TLD, true>(a, stream, dbg);
}
#endif

int configure_fwd_w4() {
    return set_attr_w4<Bf16Traits, 128>() | set_attr_w4<F16Traits, 128>() | set_attr_w4<Bf16Traits, 64>() | set_attr_w4<F16Traits, 64>();
}

}  // namespace aule_hip
