// fa_bwd_dkv4_gfx950.hip -- dK / dV of the FlashAttention-2 backward (16-bit I/O, D = 128 and, since round 4, D = 64), ONE WAVE PER SIMD.
//
// Replaces the dK/dV half of python/aule/triton_flash_amd.py:247-351 (_flash_attn_bwd_amd) where it applies; everything else
// stays on fa_bwd_gfx950.hip's kernel (8 waves x 32 keys, two waves per SIMD, lock-step: 44 % MFMA-busy, its tile period set by
// a 940-cycle barrier arrival spread and dependent issue -- DESIGN.md 3.4).  Here
//
//   * workgroup = 4 waves x 32 key rows = a 128-key KV block (causal: the pair (i, n-1-i)); a wave owns the whole 512-register
//     file: dV^T and dK^T (128 accumulator registers), its K and V fragments (64: V no longer goes through an LDS slab), the
//     row-major fragments of the query block in flight (64) in the accumulator file; scores, weights, transposed fragments,
//     L' and delta in arch VGPRs -- all named literally by fa_bwd_dkv4_asm.inc (generated: tools/gen_bw4.py, map in its docstring);
//   * the wave walks a STREAM of 32-row query blocks -- every query head of the GQA group, every block that sees the KV block --
//     software-pipelined: iteration i = [S, dP of block i+1 | arithmetic of block i | 16 of its 32 transpose reads] barrier [dV, dK
//     of block i | the other 16 transpose reads | row-major reads of block i+2 | L' of block i+2, - delta of block i+3 | LDS-DMA
//     requests of block i+4];
//   * query blocks arrive by LDS-DMA, ONE image per tensor (Q, dO: [q/4][d/16][4][16] sub-tiles with per-row-group pads, which
//     serve the ds_read_b128 fragments and the transpose reads conflict-free: tools/gen_bw4.py) into an 8-slot ring (136 KB),
//     requested four iterations ahead; the phase boundary waits with a counted vmcnt;
//   * L' = LSE log2(e) and - delta come prepared from the dQ kernel's workspace; - delta is the C operand of dP's first MFMA;
//   * one wave per SIMD pays ~4.6 cycles of issue for EVERY instruction: one stream cursor, one compare for the mask decision, no
//     range selects (the scalar offset is range-checked), the last iteration peeled;
//   * 128-key blocks make twice as many work items as the predecessor's 256-key blocks: at C3 (B4, 32q/8kv, S2048) the paired
//     causal grid is exactly 256 workgroups with the whole GQA group inside each -- no head split, no fp32 partials, no reduce
//     kernel.
//
// D = 64 (round 4): the same stream with half the MFMAs per block (16) against the same 16 scores per lane of arithmetic; a 1 KB
// LDS-DMA piece holds two row groups there, so the image is a per-piece chunk permutation instead of per-row-group pads
// (tools/gen_bw4.py, Cfg / chunk64), 128 accumulator registers, 72 arch VGPRs left to hipcc, a 68 KB ring.
//
// Covers bf16 / fp16, D = 128 / 64, causal (coff >= 0; round 5: with a sliding window too) and non-causal; deterministic (no atomics).
#include <cstdlib>
#include <type_traits>

#include "fa_device.h"
#include "fa_kernels.h"
#include "fa_fwd_tile.h"

namespace aule_hip {
namespace {

#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
#ifdef BW4_ASM_INC          // timing variants (tools/sessions/r5_s2.sh)
#include BW4_ASM_INC
#else
#include "fa_bwd_dkv4_asm.inc"
#endif

struct Dkv4Params {
    const void* q;
    const void* k;
    const void* v;
    const void* dout;
    const float* lse;
    const float* delta;
    void* dk;
    void* dv;
    int B, Hq, Hkv, Sq, Sk;
    float c;       // scale * log2(e) (sign kept: no maximum is taken here)
    float scale;   // applied to dK at the end
    int nblk;      // work items per (batch, kv head): KV blocks, or pairs of them (causal)
    int coff;      // causal position offset (query i sits at position i + coff)
    unsigned long long* dbg;   // timeline build: {iterations, cycles of [phase 1 + boundary], cycles of [phase 2]} of workgroup 0's waves
    int window;    // sliding window (round 5): key j visible to query i only if (i + coff) - j < window (0: off); never with SPILL
    char* ds;      // SPILL instances (the 5-matmul backward, fa_bwd_dqs_gfx950.hip): the dS workspace, layout in fa_kernels.h (DsLayout)
    int nq32, nkb32p;
};

constexpr int kKvBlock4 = 128;   // 4 waves x 32 key rows
constexpr int kQB = 32;          // query rows per block of the stream
constexpr int kRing4 = 8;        // slots of the LDS ring (17 KB each).  Blocks are requested four iterations ahead; eight slots keep a
                                 // block's images readable through the phase 2 that requests block i + 4 (its transpose reads are
                                 // split over both phases)

template <int N>
__device__ __forceinline__ float dkv4_acc_read() {
    float x = 0.f;
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("v_accvgpr_read_b32 %0, a%c1" : "=v"(x) : "n"(N));
#endif
    return x;
}

// accumulator block BASE + 16 d .. of the wave's key row -> the row's d = 32 d + 8 g + 4 hi .. + 3 (8-byte stores); NI = 4 (D / 32)
template <class T, int BASE, int NI, int I = 0>
__device__ __forceinline__ void dkv4_store_rows(char* row, int hi, float sc) {
    if constexpr (I < NI) {
        constexpr int d = I / 4, g4 = I % 4, N = BASE + 16 * d + 4 * g4;
        u32x2_t u;
        u[0] = T::pack2(dkv4_acc_read<N>() * sc, dkv4_acc_read<N + 1>() * sc);
        u[1] = T::pack2(dkv4_acc_read<N + 2>() * sc, dkv4_acc_read<N + 3>() * sc);
        *reinterpret_cast<u32x2_t*>(row + (32 * d + 8 * g4 + 4 * hi) * 2) = u;
        dkv4_store_rows<T, BASE, NI, I + 1>(row, hi, sc);
    }
}

__device__ __forceinline__ int dkv4_rfl(int x) { return __builtin_amdgcn_readfirstlane(x); }

// NKB (round 6): 32-key blocks per wave.  1: the stream above.  2 (D = 64 only, Bw4Asm2): a wave owns 64 keys, the workgroup a 256-key KV block;
// every fragment of the query block feeds two MFMAs (tools/gen_bw4.py, Cfg2: twice the MFMAs per iteration for the same LDS reads, scalar
// loads, requests, barrier and compiler scalars -- the D = 64 stream is issue-bound, profiles/r5_bwd_d64_ablation.txt).
template <class T, int D, int NKB>
struct Dkv4Streams { using type = Bw4Asm<T, D>; };
template <class T>
struct Dkv4Streams<T, 64, 2> { using type = Bw4Asm2<T>; };

template <class T, int D, bool CAUSAL, bool TL, bool SPILL, int NKB = 1>
__device__ __forceinline__ void dkv4_body(const Dkv4Params& p) {
    using A = typename Dkv4Streams<T, D, NKB>::type;
    using std::integral_constant;
    static_assert(NKB == 1 || (NKB == 2 && D == 64 && !SPILL && !TL), "two key blocks per wave: D = 64, recompute mode, no timeline instance");
    constexpr int KW = 32 * NKB;           // keys per wave
    constexpr int KB = 4 * KW;             // keys per workgroup (kKvBlock4 with one block per wave)
    constexpr int RB = 2 * D;
    constexpr int SLOT = A::SLOT;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
    const int wave = dkv4_rfl(tid >> 6);
#if defined(__HIP_DEVICE_COMPILE__)
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
#else
    const unsigned lds0 = 0;
#endif
    // The stream keeps ~90 scalars live in the causal SPILL instances; the tensor pointers and the dS workspace's geometry are needed only where a
    // part starts and ends, so those places read them through an opaque pointer to the kernel-argument segment (a fresh s_load each time) instead of
    // holding 20 registers for the whole kernel -- the forward's idiom (fa_fwd_w4_gfx950.hip); with them resident the allocator spilled scalars
    // into vector lanes inside the loop (tests/test_w4_audit.py forbids that).
#if defined(__HIP_DEVICE_COMPILE__)
    typedef const __attribute__((address_space(4))) Dkv4Params* KernargPtr;   // (constant address space: scalar loads)
#else
    typedef const Dkv4Params* KernargPtr;
#endif
    auto P = [&]() __attribute__((always_inline)) -> KernargPtr {
#if defined(__HIP_DEVICE_COMPILE__)
        const unsigned long long a = (unsigned long long)(uintptr_t)__builtin_amdgcn_kernarg_segment_ptr();   // (the parameter block is the kernel's only argument: offset 0)
#else
        const unsigned long long a = 0;
#endif
        unsigned lo = (unsigned)a, hi = (unsigned)(a >> 32);
        asm volatile("" : "+s"(lo), "+s"(hi));   // (not hoistable; its results count as divergent, hence the readfirstlanes)
        lo = (unsigned)dkv4_rfl((int)lo);
        hi = (unsigned)dkv4_rfl((int)hi);
        return (KernargPtr)(uintptr_t)(((unsigned long long)hi << 32) | lo);
    };
    const int g = p.Hq / p.Hkv;
    const int Sq = p.Sq, Sk = p.Sk, coff = p.coff;
    const float c = p.c;
    const int nkb = (Sk + KB - 1) / KB;
    const WorkItem w = decode_work(blockIdx.x, p.B, p.Hkv, p.Hkv, p.nblk, false);
    const size_t kvbase = (size_t)(w.b * p.Hkv + w.hk) * Sk;

    // lane constants
    unsigned tr_off, a_sub, a_sub1 = 0, vost[2] = {0, 0}, wave_pb;
    if constexpr (D == 128) {
        // one image per tensor and block (layout: tools/gen_bw4.py, Cfg.PBASE): row group rg = row / 4 is a 1024-byte piece of eight
        // [4 rows][16 d] sub-tiles at pbase(rg)
        auto pbase = [](int rg) { return 1024 * rg + (rg & 1) * 16 + ((rg >> 1) & 1) * 128 + (rg >> 2) * 256; };
        static_assert(D != 128 || (A::PB1 == 1040 && A::PB2 == 2048 + 128 && A::PB4 == 4096 + 256), "piece bases of the generator");
        tr_off = (unsigned)(hi * 1040 + ((lane >> 4) & 1) * 128 + (lane & 15) * 8);   // + the read's row-octet / d-slice immediate
        a_sub = (unsigned)(pbase(l31 >> 2) + (l31 & 3) * 32 + hi * 16);               // row l31, d = 16 ks + 8 hi ..: + 128 ks
        // per-lane source offsets of this wave's two pieces (row groups 2 w, 2 w + 1) of an image: LDS position = lane
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int rg = 2 * wave + h;
            vost[h] = (unsigned)((rg * 4 + ((lane >> 1) & 3)) * RB + ((lane >> 3) * 2 + (lane & 1)) * 16);
        }
        wave_pb = (unsigned)pbase(2 * wave);
    } else {
        // D = 64: piece p = rows 8 p .. 8 p + 7 at 1040 p; chunk (rgl, d, b, rr, h) -- row 4 rgl + rr of the piece, columns
        // 32 d + 16 b + 8 h .. + 7 -- at 16-byte position 32 d + chunk(rgl, b, rr, h) (tools/gen_bw4.py, chunk64)
        static_assert(D != 64 || (A::PB1 == 1040 && A::PB2 == 2080), "piece bases of the generator");
        auto chunk = [](int rgl, int b, int rr, int h) { return 16 * rgl + 8 * (rgl ^ b) + 2 * rr + (h ^ b); };
        // transpose read: lane -> row group hi of the piece, sub-tile b = lane bit 4, row (lane >> 2) & 3, 8 bytes (lane & 3) of the 32
        tr_off = (unsigned)(16 * chunk(hi, (lane >> 4) & 1, (lane >> 2) & 3, (lane >> 1) & 1) + (lane & 1) * 8);   // + piece / d-slice immediate
        // row-major fragment of row l31, k-slice ks = 2 d + b (columns 16 ks + 8 hi ..): base of b, + 512 d
        a_sub = (unsigned)(1040 * (l31 >> 3) + 16 * chunk((l31 >> 2) & 1, 0, l31 & 3, hi));
        a_sub1 = (unsigned)(1040 * (l31 >> 3) + 16 * chunk((l31 >> 2) & 1, 1, l31 & 3, hi));
        // source offset of this wave's piece (piece w of an image): lane l fills chunk l
        const int cd = lane >> 5, g5 = lane & 31, rgl = g5 >> 4, cb = rgl ^ ((g5 >> 3) & 1), rr = (g5 >> 1) & 3, ch = (g5 & 1) ^ cb;
        vost[0] = (unsigned)((wave * 8 + 4 * rgl + rr) * RB + (2 * cd + cb) * 32 + ch * 16);
        wave_pb = (unsigned)(1040 * wave);
    }
    // (the LDS base goes into the per-lane / per-wave constants once: a relocation the compiler cannot fold costs an s_add per use in the loop)
    tr_off += lds0; a_sub += lds0; a_sub1 += lds0; wave_pb += lds0;
    const unsigned lvo = (unsigned)(hi * 16);   // L' / delta: rows 8 g + 4 hi .. + 3 of the block per dwordx4
    const unsigned svo = (unsigned)(lane * 16);   // SPILL: the lane's 16 bytes of a dS unit's k-step (unit = [kk][lane][16 B])

    unsigned long long tl_a = 0, tl_b = 0, tl_n = 0, tl_w = 0;
    const int nparts = (CAUSAL && (nkb - 1 - w.blk) != w.blk) ? 2 : 1;
    for (int part = 0; part < nparts; ++part) {
        // Causal pair: the EARLY key block (the long stream: every query block from its diagonal to the end) first, walked DOWNWARDS from the
        // last query block; then the late one (the short stream) upwards.  Every work item of a (batch, kv head) unit then starts at the
        // same query block -- the last -- and, the pairs being balanced, enters its second part reading the block the others read too
        // (block x / g - 4 of the sweep): the unit's items run in lock step through BOTH parts and an XCD's L2 serves all but one of them
        // (round 6; see the cursor below for the measurements).
        const int kb = CAUSAL ? (part == 0 ? w.blk : nkb - 1 - w.blk) : w.blk;
        const int down = dkv4_rfl((CAUSAL && part == 0) ? 1 : 0);
        const int n0w = kb * KB + wave * KW;
        const int kvrow = n0w + l31;
        {
            const __amdgpu_buffer_rsrc_t krs = make_srd(reinterpret_cast<const char*>(P()->k) + kvbase * RB, (unsigned)Sk * RB);
            const __amdgpu_buffer_rsrc_t vrs = make_srd(reinterpret_cast<const char*>(P()->v) + kvbase * RB, (unsigned)Sk * RB);
            if constexpr (NKB == 2) A::load_kv(krs, vrs, (unsigned)(kvrow * RB + hi * 16), (unsigned)((kvrow + 32) * RB + hi * 16));
            else A::load_kv(krs, vrs, (unsigned)(kvrow * RB + hi * 16));
        }
        A::zero_acc();

        const int W = SPILL ? 0 : P()->window;   // (the 5-matmul mode never carries a window: the dispatcher keeps windowed problems on the recompute pair)
        int nq32 = (Sq + kQB - 1) / kQB;
        // (window: the last query that sees the block's last key kb 128 + 127 sits at position key + W - 1: the stream of a KV block ends there)
        if (W > 0) nq32 = min(nq32, max(0, kb * KB + KB - 1 + W - 1 - coff) / kQB + 1);
        const int first_qt = CAUSAL ? max(0, kb * KB - coff) / kQB : 0;
        const int ntq = nq32 > first_qt ? nq32 - first_qt : 0;
        const int nit = ntq * g;   // flattened (query block, query head of the group) stream

        // block x of the stream: query block first_qt + x / g of head x % g of the group (round 6; see the cursor) -- kept as cursors that advance with
        // the stream: (blocks left in the head, global row of the block inside the group's rows).  ONE descriptor per tensor
        // for the whole group, the head and the block go into the request's scalar offset: per-head descriptors cost ~90 scalar
        // instructions per iteration between the MFMA statements.  (A ragged last block of a head then reads the first rows
        // of the NEXT head of the group instead of zeros; its weights are masked to exactly 0, and those rows belong to the
        // same dK / dV sum anyway.  Beyond the group's last head the descriptor's bounds return 0.)
        const size_t grows = (size_t)(w.b * p.Hq + w.hk * g) * Sq;   // first row of the group's first head
        __amdgpu_buffer_rsrc_t qrs = make_srd(reinterpret_cast<const char*>(P()->q) + grows * RB, (unsigned)dkv4_rfl(g * Sq * RB));
        __amdgpu_buffer_rsrc_t grs = make_srd(reinterpret_cast<const char*>(P()->dout) + grows * RB, (unsigned)dkv4_rfl(g * Sq * RB));
        __amdgpu_buffer_rsrc_t lrs = make_srd(P()->lse + grows, (unsigned)dkv4_rfl(g * Sq * 4));
        __amdgpu_buffer_rsrc_t drs = make_srd(P()->delta + grows, (unsigned)dkv4_rfl(g * Sq * 4));
#if defined(__HIP_DEVICE_COMPILE__)
        // (whole descriptors, not words: the pairs share their size / flag words, and the compiler would re-assemble a four-register
        // tuple from the shared words in front of every statement that takes one -- four s_mov per iteration)
        asm volatile("" : "+s"(qrs), "+s"(grs), "+s"(lrs), "+s"(drs));
#endif
        // Round 6: the stream is BLOCK-major, head-minor -- query block t of head 0, of head 1, .. of head g - 1, then block t + 1 -- where it
        // used to be head-major.  Why: the work items of one (batch, kv head) unit run concurrently on one XCD and read the same Q / dO rows.
        // Key block kb's stream starts at block 4 kb of a head; walking the heads one after the other, an item with a short per-head stream
        // (a late key block) is in head 1 while its neighbours are still in head 0, and at C3 (4 heads per group) an XCD's 32 items were
        // spread over 16 MB of Q / dO against a 4 MB L2: 553 MB fetched for 170 MB of inputs (profiles/r5_fwdbwd_c3_*).  Head-minor, item kb
        // reads block 4 kb + x / g at iteration x: neighbours follow each other 4 g iterations apart (a few hundred KB of reuse distance),
        // and the second parts of all pairs of a unit run in lock step (they all start at block x / g - 4).  MHA (g = 1) is the same stream
        // as before.
        // Measured at C3 (B4 32q/8kv S2048, 170 MB of inputs; FETCH_SIZE of this kernel, profiles/r6_bwd_dkv_order.txt): head-major 578 MB; head-minor
        // upwards 477 MB (neighbours still 4 g = 16 iterations apart: the XCD's 32 items push 8 MB through the 4 MB L2 in between); lock step (the
        // directions below) 233 MB.  Time: neutral -- the re-reads came from the Infinity Cache.
        struct Cur { int hleft, left, row; };   // heads left at this block (this one included); the block's place in its head is ntq - left; row of the block in the group
        // upwards: head 0 .. g - 1 of block t, then block t + 1; downwards: head g - 1 .. 0 of block t, then block t - 1 (the mirror image,
        // so that the SPILL unit address stays linear in the stream position)
        const int row_first = down ? (g - 1) * Sq + (first_qt + ntq - 1) * kQB : first_qt * kQB;
        const int left_first = down ? 1 : ntq;
        const int row_hstep = down ? -Sq : Sq;                                   // to the next head of the same block
        const int row_wrap = down ? (g - 1) * Sq - kQB : kQB - (g - 1) * Sq;     // from the last head of a block to the first head of the next block
        const int left_step = down ? 1 : -1;
        auto adv = [&](Cur& cu) __attribute__((always_inline)) {
            if (--cu.hleft == 0) { cu.hleft = g; cu.left += left_step; cu.row += row_wrap; }
            else cu.row += row_hstep;
        };
        // (a cursor behind the stream's last block needs no special offset: the scalar offset is part of the descriptor's range
        // check on gfx950 -- tools/probe_soffset.hip -- so rows >= g Sq read zeros / the request writes zeros)
        auto slot_lds = [&](int x) __attribute__((always_inline)) { return (unsigned)(x & (kRing4 - 1)) * SLOT; };   // (+ tr_off / a_sub / wave_pb: those carry the LDS base)
        // Does anybody's lane need a mask in block t of a head?  The causal diagonal covers the head's first t_diag blocks (every
        // block if the wave's 32 keys run past Sk), a ragged Sq its last one.
        const int diag_x = CAUSAL ? n0w + KW - 1 - coff - first_qt * kQB : 0;   // block t crosses the diagonal (of the wave's last key) iff t kQB < diag_x
        const int t_diag = (n0w + KW > Sk) ? 0x7fffffff : (diag_x > 0 ? (diag_x + kQB - 1) / kQB : 0);
        int t_plain_end = (Sq % kQB) != 0 && nq32 * kQB > Sq ? ntq - 1 : 0x7fffffff;   // blocks t_diag <= t < t_plain_end need no mask:
        // (window: the first query row that does NOT see the wave's first key n0w is n0w + W - coff; blocks that reach it need the mask)
        if (W > 0) t_plain_end = min(t_plain_end, max(0, max(0, n0w + W - coff) / kQB - first_qt));
        // ONE unsigned compare per iteration, t - t_diag < t_span, on the cursor's `left` = ntq - t: (ntq - t_diag) - left < t_span
        const unsigned t_lo = (unsigned)dkv4_rfl(ntq - t_diag);
        const unsigned t_span = (unsigned)dkv4_rfl(t_plain_end > t_diag ? t_plain_end - t_diag : 0);
        // mask of block t for this lane: rows [lo, lo + wd) of the block are valid (as crow(r) + 4 hi)
        auto mask_of = [&](int t, int& lo, int& wd, int kofs = 0) __attribute__((always_inline)) {   // kofs: 32 for the wave's second key block (NKB = 2)
            const int q0 = (first_qt + t) * kQB;
            int lane_o = lane;
            asm volatile("" : "+v"(lane_o));
            const int kr = n0w + kofs + (lane_o & 31);
            const int lo_r = CAUSAL ? max(0, kr - coff - q0) : 0;
            int hi_r = min(kQB, Sq - q0);
            if (W > 0) hi_r = min(hi_r, kr + W - coff - q0);   // q + coff - kr < W
            lo = lo_r - 4 * (lane_o >> 5);
            wd = (kr < Sk && hi_r > lo_r) ? hi_r - lo_r : 0;
        };

        // SPILL: the block (query block qb32, head hh of the group) of this wave's 32 keys (32-key block kb32 = 4 kb + wave) is unit
        // x = g qb32 + hh of column (group, kb32) (fa_kernels.h, DsLayout: block-major, head-minor like the stream -- round 6): stream
        // position i is unit g first_qt + i walking upwards and g (first_qt + ntq) - 1 - i walking downwards -- a running offset (sp_off)
        // that moves by one unit per iteration, no cursor
        const __amdgpu_buffer_rsrc_t srs = [&]() __attribute__((always_inline)) {
            if constexpr (SPILL) {
                const long long xs = (long long)g * P()->nq32;
                const long long col = (long long)(w.b * P()->Hkv + w.hk) * P()->nkb32p + (kb * 4 + wave);
                return make_srd(P()->ds + ((col * xs + (long long)g * first_qt) << 11), (unsigned)dkv4_rfl((int)((xs - (long long)g * first_qt) << 11)));
            } else {
                return make_srd(nullptr, 0);
            }
        }();
        if (nit > 0) {
            // ---- stream start: blocks 0 .. 3 requested, L' / delta of blocks 0 and 1, the fragments of block 0, S_0 / dP_0.
            // ONE cursor walks the stream, four blocks ahead of the iteration (the block being requested); what an iteration needs
            // of blocks i (its place in the head, for the mask) and i + 2 (its row, for L' / delta) are values the cursor had four /
            // two iterations earlier, kept by block parity (the loop is unrolled by two) -- a second and a third cursor cost ten
            // scalar instructions per iteration, and one wave per SIMD pays ~4.6 cycles of issue for every instruction.
            Cur c4{g, left_first, row_first};
            int t_cur[2], t_nxt[2], row_nxt[2], row_01[2];   // (t_*: the cursor's `left` at that block)
            unsigned sp_off = (SPILL && down) ? (unsigned)(g * ntq - 1) << 11 : 0u;   // SPILL: byte offset of the iteration's unit behind the stream's first
            const unsigned sp_step = (SPILL && down) ? (unsigned)-2048 : 2048u;
#pragma unroll
            for (int x = 0; x < 4; ++x) {
                if constexpr (NKB == 2) A::dma_block(slot_lds(x) + wave_pb, qrs, grs, (unsigned)c4.row * (unsigned)RB, vost[0]);
                else A::dma_block(slot_lds(x) + wave_pb, qrs, grs, (unsigned)c4.row * (unsigned)RB, vost[0], vost[1]);
                if (x < 2) { t_cur[x] = c4.left; row_01[x] = c4.row; }
                else { t_nxt[x - 2] = c4.left; row_nxt[x - 2] = c4.row; }
                adv(c4);
            }
            if constexpr (NKB == 2) {
                // (one - delta buffer: block 0's goes straight in, block 1's is parked in X until dP_0's first MFMAs have read block 0's)
                A::template load_scal<0, 1>(lrs, drs, lvo, (unsigned)row_01[0] * 4u);
                A::template load_scal<1, 0>(lrs, drs, lvo, (unsigned)row_01[1] * 4u);
                A::load_delta_x(drs, lvo, (unsigned)row_01[1] * 4u);
            } else {
                A::template load_scal<0, 0>(lrs, drs, lvo, (unsigned)row_01[0] * 4u);
                A::template load_scal<1, 0>(lrs, drs, lvo, (unsigned)row_01[1] * 4u);
                A::load_delta_x(drs, lvo, (unsigned)row_nxt[0] * 4u);   // - delta of block 2: parked until dP_0 has read block 0's
            }
            asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
            const __amdgpu_buffer_rsrc_t nosrd = make_srd(nullptr, 0);
            auto rm_reads = [&](int x) __attribute__((always_inline)) {   // row-major fragments of block x -> the accumulator file
                const unsigned b = slot_lds(x) + a_sub, b1 = slot_lds(x) + a_sub1;
                if constexpr (NKB == 2) {
                    A::template p2<0, 0, 0, 1, 0, 0, 0, 0>(0.f, 0, 0, b, b1, nosrd, nosrd, 0, 0, 0, nosrd, nosrd, 0, 0);
                    A::template p2<1, 0, 0, 1, 0, 0, 0, 0>(0.f, 0, 0, b, b1, nosrd, nosrd, 0, 0, 0, nosrd, nosrd, 0, 0);
                    A::template p2<2, 0, 0, 1, 0, 0, 0, 0>(0.f, 0, 0, b, b1, nosrd, nosrd, 0, 0, 0, nosrd, nosrd, 0, 0);
                    A::template p2<3, 0, 0, 1, 0, 0, 0, 0>(0.f, 0, 0, b, b1, nosrd, nosrd, 0, 0, 0, nosrd, nosrd, 0, 0);
                } else {
                    A::template p2<0, 0, 0, 1, 0, 0>(b, b1, 0, nosrd, nosrd, 0, 0, 0, 0, nosrd, nosrd, 0, 0, 0);
                    A::template p2<1, 0, 0, 1, 0, 0>(b, b1, 0, nosrd, nosrd, 0, 0, 0, 0, nosrd, nosrd, 0, 0, 0);
                    A::template p2<2, 0, 0, 1, 0, 0>(b, b1, 0, nosrd, nosrd, 0, 0, 0, 0, nosrd, nosrd, 0, 0, 0);
                    A::template p2<3, 0, 0, 1, 0, 0>(b, b1, 0, nosrd, nosrd, 0, 0, 0, 0, nosrd, nosrd, 0, 0, 0);
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            };
            rm_reads(0);
            if constexpr (NKB == 2) {   // S_0, then dP_0 of both key blocks (starts from - delta of block 0)
                A::template p1<0, 1, 1, 0, 0>(c, 0, 0, 0, 0, 0);
                A::template p1<1, 1, 1, 0, 0>(c, 0, 0, 0, 0, 0);
                A::template p1<2, 1, 1, 0, 0>(c, 0, 0, 0, 0, 0);
                A::template p1<3, 1, 1, 0, 0>(c, 0, 0, 0, 0, 0);
            } else {
                A::template p1<0, 1, 1, 0, 0>(c, 0, 0, 0);   // S_0, dP_0 (parity 0 buffers)
                A::template p1<1, 1, 1, 0, 0>(c, 0, 0, 0);
                A::template p1<2, 1, 1, 0, 0>(c, 0, 0, 0);
                A::template p1<3, 1, 1, 0, 0>(c, 0, 0, 0);
            }
            if constexpr (NKB == 2) {
                A::template p2<0, 0, 0, 0, 0, 0, 1, 0>(0.f, 0, 0, 0, 0, nosrd, nosrd, 0, 0, 0, nosrd, nosrd, 0, 0);
                A::template p2<1, 0, 0, 0, 0, 0, 1, 0>(0.f, 0, 0, 0, 0, nosrd, nosrd, 0, 0, 0, nosrd, nosrd, 0, 0);
                A::template p2<2, 0, 0, 0, 0, 0, 1, 0>(0.f, 0, 0, 0, 0, nosrd, nosrd, 0, 0, 0, nosrd, nosrd, 0, 0);
                A::template p2<3, 0, 0, 0, 0, 0, 1, 0>(0.f, 0, 0, 0, 0, nosrd, nosrd, 0, 0, 0, nosrd, nosrd, 0, 0);
            }
            A::mov_delta_x();
            if (nit > 1) rm_reads(1);

            // ---- the stream.  QK = 0: the last iteration (no next block to start)
            auto iteration = [&](auto par_tag, auto qk_tag, int i) __attribute__((always_inline)) {
                constexpr int PAR = decltype(par_tag)::value, QK = decltype(qk_tag)::value;
                unsigned long long t0 = 0;
                if constexpr (TL) t0 = __builtin_amdgcn_s_memtime();
                const int left = t_cur[PAR];   // block t = ntq - left of its head
                const unsigned trb = slot_lds(i) + tr_off;
#define DKV4_P1(AR, LO, WD)                               \
    A::template p1<0, PAR, QK, AR, 1>(c, LO, WD, trb);     \
    A::template p1<1, PAR, QK, AR, 1>(c, LO, WD, trb);     \
    A::template p1<2, PAR, QK, AR, 1>(c, LO, WD, trb);     \
    A::template p1<3, PAR, QK, AR, 1>(c, LO, WD, trb);
#define DKV4_P1X(AR, LA, WA, LB, WB)                              \
    A::template p1<0, PAR, QK, AR, 1>(c, LA, WA, LB, WB, trb);     \
    A::template p1<1, PAR, QK, AR, 1>(c, LA, WA, LB, WB, trb);     \
    A::template p1<2, PAR, QK, AR, 1>(c, LA, WA, LB, WB, trb);     \
    A::template p1<3, PAR, QK, AR, 1>(c, LA, WA, LB, WB, trb);
                // (NKB = 2: the mask of the wave's second key block rides along; a SPLIT build -- tools/gen_bw4.py BW4_K2_SPLIT, an experiment --
                // applies it in phase 2, where that block's arithmetic then sits)
                int lob = 0, wdb = 0;
                bool masked = false;
                if (__builtin_expect(t_lo - (unsigned)left < t_span, 1)) {   // (expected: the masked statements then sit outside the loop body)
                    if constexpr (NKB == 2) { DKV4_P1X(1, 0, 0, 0, 0) } else { DKV4_P1(1, 0, 0) }
                } else {
                    int lo, wd;
                    mask_of(ntq - left, lo, wd);
                    if constexpr (NKB == 2) {
                        mask_of(ntq - left, lob, wdb, 32);
                        masked = true;
                        DKV4_P1X(2, lo, wd, lob, wdb)
                    } else {
                        DKV4_P1(2, lo, wd)
                    }
                }
#undef DKV4_P1
#undef DKV4_P1X
                // block i + 2 has landed for everybody (all but this wave's newest NP requests -- block i + 3 -- are complete:
                // the scalars of block i + 1 among them)
                unsigned long long tw = 0;
                if constexpr (TL) tw = __builtin_amdgcn_s_memtime();
                // (SPILL with the dS stores behind the DMA pieces: the two stores of the previous iteration may stay out as well)
                asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(A::NP + (SPILL ? A::ST_LATE : 0)) : "memory");
                unsigned long long t1 = 0;
                if constexpr (TL) { t1 = __builtin_amdgcn_s_memtime(); tl_a += t1 - t0; tl_w += t1 - tw; }
                const unsigned b = slot_lds(i + 2) + a_sub, b1 = slot_lds(i + 2) + a_sub1;
                if constexpr (SPILL && A::ST_LATE == 0) A::store_ds(srs, svo, sp_off);
                {
                    const unsigned lso = (unsigned)row_nxt[PAR] * 4u, lso3 = (unsigned)row_nxt[PAR ^ 1] * 4u, dso = (unsigned)c4.row * (unsigned)RB;
                    const unsigned dl = slot_lds(i + 4) + wave_pb;
                    if constexpr (NKB == 2) {
                        (void)lso3;   // (one - delta buffer: - delta of block i + 2 rides with its L')
#define DKV4_P2X(AR)                                                                                            \
    A::template p2<0, PAR, 1, 1, 1, 1, QK, AR>(c, lob, wdb, b, b1, lrs, drs, lvo, lso, dl, qrs, grs, dso, vost[0]);   \
    A::template p2<1, PAR, 1, 1, 1, 1, QK, AR>(c, lob, wdb, b, b1, lrs, drs, lvo, lso, dl, qrs, grs, dso, vost[0]);   \
    A::template p2<2, PAR, 1, 1, 1, 1, QK, AR>(c, lob, wdb, b, b1, lrs, drs, lvo, lso, dl, qrs, grs, dso, vost[0]);   \
    A::template p2<3, PAR, 1, 1, 1, 1, QK, AR>(c, lob, wdb, b, b1, lrs, drs, lvo, lso, dl, qrs, grs, dso, vost[0]);
                        if constexpr (A::SPLIT != 0) {
                            if (__builtin_expect(!masked, 1)) { DKV4_P2X(1) } else { DKV4_P2X(2) }
                        } else {
                            (void)masked;
                            DKV4_P2X(1)
                        }
#undef DKV4_P2X
                    } else {
                        A::template p2<0, PAR, 1, 1, 1, 1>(b, b1, trb, lrs, drs, lvo, lso, lso3, dl, qrs, grs, dso, vost[0], vost[1]);
                        A::template p2<1, PAR, 1, 1, 1, 1>(b, b1, trb, lrs, drs, lvo, lso, lso3, dl, qrs, grs, dso, vost[0], vost[1]);
                        A::template p2<2, PAR, 1, 1, 1, 1>(b, b1, trb, lrs, drs, lvo, lso, lso3, dl, qrs, grs, dso, vost[0], vost[1]);
                        A::template p2<3, PAR, 1, 1, 1, 1>(b, b1, trb, lrs, drs, lvo, lso, lso3, dl, qrs, grs, dso, vost[0], vost[1]);
                    }
                }
                if constexpr (SPILL && A::ST_LATE != 0) A::store_ds(srs, svo, sp_off);
                if constexpr (SPILL) sp_off += sp_step;
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the fragments of block i + 2 (phase 1 of the next iteration reads them)
                t_cur[PAR] = t_nxt[PAR]; t_nxt[PAR] = c4.left; row_nxt[PAR] = c4.row;
                adv(c4);
                if constexpr (TL) { tl_b += __builtin_amdgcn_s_memtime() - t1; ++tl_n; }
            };
            using I0 = integral_constant<int, 0>;
            using I1 = integral_constant<int, 1>;
            int i = 0;
            for (; i + 2 < nit; i += 2) {
                iteration(I0{}, I1{}, i);
                iteration(I1{}, I1{}, i + 1);
            }
            if (nit - i == 2) {
                iteration(I0{}, I1{}, i);
                iteration(I1{}, I0{}, i + 1);
            } else {
                iteration(I0{}, I0{}, i);
            }
            asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");   // the out-of-range requests of the last iterations too
        }

        // ---- dK (scaled), dV of the wave's key rows
        asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");   // the last MFMAs -> v_accvgpr_read
        if (kvrow < Sk) {
            int lane_o = lane;
            asm volatile("" : "+v"(lane_o));
            const size_t row = kvbase + (size_t)(n0w + (lane_o & 31));
            if constexpr (NKB == 2) {   // accumulator file: dV_A a[0:31], dV_B a[32:63], dK_A a[64:95], dK_B a[96:127]
                dkv4_store_rows<T, 0, D / 8>(reinterpret_cast<char*>(P()->dv) + row * RB, lane_o >> 5, 1.0f);
                dkv4_store_rows<T, 64, D / 8>(reinterpret_cast<char*>(P()->dk) + row * RB, lane_o >> 5, P()->scale);
            } else {
                dkv4_store_rows<T, 0, D / 8>(reinterpret_cast<char*>(P()->dv) + row * RB, lane_o >> 5, 1.0f);
                dkv4_store_rows<T, D / 2, D / 8>(reinterpret_cast<char*>(P()->dk) + row * RB, lane_o >> 5, P()->scale);
            }
        }
        if constexpr (NKB == 2) {
            if (kvrow + 32 < Sk) {
                int lane_o = lane;
                asm volatile("" : "+v"(lane_o));
                const size_t row = kvbase + (size_t)(n0w + 32 + (lane_o & 31));
                dkv4_store_rows<T, 32, D / 8>(reinterpret_cast<char*>(P()->dv) + row * RB, lane_o >> 5, 1.0f);
                dkv4_store_rows<T, 96, D / 8>(reinterpret_cast<char*>(P()->dk) + row * RB, lane_o >> 5, P()->scale);
            }
        }
        __syncthreads();
    }
    if constexpr (TL) {
        if (blockIdx.x == 0 && lane == 0) {
            p.dbg[wave * 4 + 0] = tl_n; p.dbg[wave * 4 + 1] = tl_a; p.dbg[wave * 4 + 2] = tl_b; p.dbg[wave * 4 + 3] = tl_w;   // (tl_w: the boundary's wait + barrier, part of tl_a)
        }
    }
}

template <class T, bool CAUSAL, bool TL = false, bool SPILL = false>
__global__ void __launch_bounds__(256, 1) __attribute__((amdgpu_num_vgpr(40))) fa_bwd_dkv4_kernel(const Dkv4Params p) {
    static_assert(Bw4Asm<T, 128>::NV == 40, "amdgpu_num_vgpr must be the generator's NV");
    dkv4_body<T, 128, CAUSAL, TL, SPILL>(p);
}

// D = 64: 72 arch VGPRs for hipcc (the attribute takes a literal, hence a kernel of its own)
template <class T, bool CAUSAL, bool SPILL = false, bool TL = false>
__global__ void __launch_bounds__(256, 1) __attribute__((amdgpu_num_vgpr(72))) fa_bwd_dkv4_kernel_d64(const Dkv4Params p) {
    static_assert(Bw4Asm<T, 64>::NV == 72, "amdgpu_num_vgpr must be the generator's NV");
    dkv4_body<T, 64, CAUSAL, TL, SPILL>(p);
}

// D = 64, two 32-key blocks per wave (round 6): 40 arch VGPRs for hipcc like D = 128
template <class T, bool CAUSAL>
__global__ void __launch_bounds__(256, 1) __attribute__((amdgpu_num_vgpr(40))) fa_bwd_dkv4_kernel_d64k2(const Dkv4Params p) {
    static_assert(Bw4Asm2<T>::NV == 40, "amdgpu_num_vgpr must be the generator's NV");
    dkv4_body<T, 64, CAUSAL, false, false, 2>(p);
}

#pragma clang diagnostic pop

template <int D>
constexpr int kDkv4Lds = kRing4 * Bw4Asm<Bf16Traits, D>::SLOT;

// Does the D = 64 problem take the two-key-blocks-per-wave instance?  Its work items are 256-key blocks (pairs of them): half as many, each
// ~1.32 x as long as a 128-key item (twice the MFMAs in ~0.66 of twice the time: profiles/r6_bwd_d64_k2.txt).  Whole rounds of the chip decide:
// ceil(items / CUs) of either kind, priced.  AULE_HIP_BWD_DKV_K2=0 / 1 pins it (A/B, tests).
inline long long dkv4_items_of(const BwdArgs& a, int kb) {
    const int nkb = (a.Sk + kb - 1) / kb;
    return (long long)a.B * a.Hkv * (a.causal ? (nkb + 1) / 2 : nkb);
}
inline bool dkv4_use_k2(const BwdArgs& a) {
    static const int mode = [] {
        const char* e = std::getenv("AULE_HIP_BWD_DKV_K2");
        return e == nullptr ? -1 : (e[0] == '0' ? 0 : 1);
    }();
    if (a.D != 64 || a.ds != nullptr || a.dbg != nullptr) return false;   // (the 5-matmul mode and the timeline instances stay on the one-block stream)
    if (mode >= 0) return mode == 1;
    const long long cus = device_cu_count(a.device);
    const long long i1 = dkv4_items_of(a, 128), i2 = dkv4_items_of(a, 256);
    const long long r1 = (i1 + cus - 1) / cus, r2 = (i2 + cus - 1) / cus;
    return r2 * 132 < r1 * 100;
}

template <class T, int D>
int launch_dkv4(const BwdArgs& a, hipStream_t stream) {
    Dkv4Params p;
    p.q = a.q; p.k = a.k; p.v = a.v; p.dout = a.dout; p.lse = a.lse2; p.delta = a.ndelta;   // (L' = LSE log2(e) and - delta, written by the dQ kernel / the delta pass behind delta)
    p.dk = a.dk; p.dv = a.dv;
    p.B = a.B; p.Hq = a.Hq; p.Hkv = a.Hkv; p.Sq = a.Sq; p.Sk = a.Sk;
    p.c = a.scale * kLog2e;
    p.scale = a.scale;
    p.coff = a.causal ? a.coff : 0;
    p.window = a.window > 0 ? a.window : 0;
    const bool k2 = D == 64 && dkv4_use_k2(a);
    const int nkb = k2 ? (a.Sk + 255) / 256 : (a.Sk + kKvBlock4 - 1) / kKvBlock4;
    p.nblk = a.causal ? (nkb + 1) / 2 : nkb;
    const dim3 grid((unsigned)(p.nblk * a.B * a.Hkv)), block(256);
    p.dbg = a.dbg;
    const DsLayout dl = DsLayout::of(a.Hq, a.Hkv, a.Sq, a.Sk);
    p.ds = reinterpret_cast<char*>(a.ds); p.nq32 = dl.nq32; p.nkb32p = dl.nkb32p;
    const bool spill = a.ds != nullptr;   // the 5-matmul backward: dS goes to the workspace for fa_bwd_dqs_gfx950.hip
    constexpr int LDS = kDkv4Lds<D>;
    if constexpr (D == 64) {
        if (k2) {
            if (a.causal)
                hipLaunchKernelGGL((fa_bwd_dkv4_kernel_d64k2<T, true>), grid, block, LDS, stream, p);
            else
                hipLaunchKernelGGL((fa_bwd_dkv4_kernel_d64k2<T, false>), grid, block, LDS, stream, p);
            return (int)hipGetLastError();
        }
#ifdef AULE_DEBUG_HOOKS
        if constexpr (std::is_same<T, Bf16Traits>::value) {
            if (a.dbg != nullptr) {   // timeline build at D = 64 (round 5: tools/timeline_dkv4.py ... 64)
                if (a.causal) {
                    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&fa_bwd_dkv4_kernel_d64<T, true, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
                    hipLaunchKernelGGL((fa_bwd_dkv4_kernel_d64<T, true, false, true>), grid, block, LDS, stream, p);
                } else {
                    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&fa_bwd_dkv4_kernel_d64<T, false, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
                    hipLaunchKernelGGL((fa_bwd_dkv4_kernel_d64<T, false, false, true>), grid, block, LDS, stream, p);
                }
                return (int)hipGetLastError();
            }
        }
#endif
        if (spill) {
            if (a.causal)
                hipLaunchKernelGGL((fa_bwd_dkv4_kernel_d64<T, true, true>), grid, block, LDS, stream, p);
            else
                hipLaunchKernelGGL((fa_bwd_dkv4_kernel_d64<T, false, true>), grid, block, LDS, stream, p);
        } else if (a.causal)
            hipLaunchKernelGGL((fa_bwd_dkv4_kernel_d64<T, true>), grid, block, LDS, stream, p);
        else
            hipLaunchKernelGGL((fa_bwd_dkv4_kernel_d64<T, false>), grid, block, LDS, stream, p);
        return (int)hipGetLastError();
    } else {
#ifdef AULE_DEBUG_HOOKS
        if constexpr (std::is_same<T, Bf16Traits>::value) {
            if (a.dbg != nullptr) {   // timeline build (tools/timeline_dkv4.py)
                if (a.causal) {
                    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&fa_bwd_dkv4_kernel<T, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
                    hipLaunchKernelGGL((fa_bwd_dkv4_kernel<T, true, true>), grid, block, LDS, stream, p);
                } else {
                    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&fa_bwd_dkv4_kernel<T, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
                    hipLaunchKernelGGL((fa_bwd_dkv4_kernel<T, false, true>), grid, block, LDS, stream, p);
                }
                return (int)hipGetLastError();
            }
        }
#endif
        if (spill) {
            if (a.causal)
                hipLaunchKernelGGL((fa_bwd_dkv4_kernel<T, true, false, true>), grid, block, LDS, stream, p);
            else
                hipLaunchKernelGGL((fa_bwd_dkv4_kernel<T, false, false, true>), grid, block, LDS, stream, p);
        } else if (a.causal)
            hipLaunchKernelGGL((fa_bwd_dkv4_kernel<T, true>), grid, block, LDS, stream, p);
        else
            hipLaunchKernelGGL((fa_bwd_dkv4_kernel<T, false>), grid, block, LDS, stream, p);
        return (int)hipGetLastError();
    }
}

}  // namespace

// Shapes the one-wave-per-SIMD dK/dV kernel CAN take: 16-bit, D = 128 or 64, no window, causal offset >= 0, a GQA group's rows inside
// one 2 GB descriptor.  Whether it is taken: bwd_dkv4_items() against the predecessor's grid, in the dispatcher (fa_bwd_gfx950.hip).
bool bwd_dkv4_applicable(const BwdArgs& a) {
    // AULE_HIP_BWD_DKV=old: the two-waves-per-SIMD kernel everywhere (A/B); =new: this kernel wherever it CAN run (tests)
    static const int mode = [] {
        const char* e = std::getenv("AULE_HIP_BWD_DKV");
        return e == nullptr ? 0 : (e[0] == 'o' ? 1 : (e[0] == 'n' ? 2 : 0));
    }();
    if (mode == 1) return false;
    if (a.dtype != kBF16 && a.dtype != kF16) return false;
    if (a.D != 128 && a.D != 64) return false;
    if (a.window > 0 && !a.causal) return false;      // (round 5: causal sliding windows run here too; a window without the causal rule stays on the predecessor)
    if (a.causal && a.coff < 0) return false;
    if (a.Hkv <= 0 || a.Hq % a.Hkv != 0) return false;
    // one descriptor covers the rows of a whole GQA group; byte offsets inside it are 32-bit
    if ((long long)(a.Hq / a.Hkv) * a.Sq * a.D * 2 >= (1LL << 31) || (long long)a.Sk * a.D * 2 >= (1LL << 31)) return false;
    return true;
}

// Work items of this kernel's grid: (batch, KV head, 128-key block -- causal: pair of blocks); the whole GQA group runs inside one.
long long bwd_dkv4_items(const BwdArgs& a) {
    const int nkb = (a.Sk + kKvBlock4 - 1) / kKvBlock4;
    return (long long)a.B * a.Hkv * (a.causal ? (nkb + 1) / 2 : nkb);
}

// D = 64: does launch_bwd_dkv4 run the two-key-blocks-per-wave instance for this problem?  (the dispatcher's route record)
bool bwd_dkv4_k2(const BwdArgs& a) { return a.D == 64 && dkv4_use_k2(a); }

// AULE_HIP_BWD_DKV=new: take every problem bwd_dkv4_applicable() accepts (tests)
bool bwd_dkv4_forced() {
    static const int v = [] {
        const char* e = std::getenv("AULE_HIP_BWD_DKV");
        return (e != nullptr && e[0] == 'n') ? 1 : 0;
    }();
    return v == 1;
}

int launch_bwd_dkv4(const BwdArgs& a, hipStream_t stream) {
    if (a.D == 128) {
        if (a.dtype == kBF16) return launch_dkv4<Bf16Traits, 128>(a, stream);
        if (a.dtype == kF16) return launch_dkv4<F16Traits, 128>(a, stream);
    } else if (a.D == 64) {
        if (a.dtype == kBF16) return launch_dkv4<Bf16Traits, 64>(a, stream);
        if (a.dtype == kF16) return launch_dkv4<F16Traits, 64>(a, stream);
    }
    return -1;
}

int configure_bwd_dkv4() {
    int rc = 0;
    auto set = [&](const void* f, int lds) { rc |= (int)hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, lds); };
    set(reinterpret_cast<const void*>(&fa_bwd_dkv4_kernel<Bf16Traits, true>), kDkv4Lds<128>);
    set(reinterpret_cast<const void*>(&fa_bwd_dkv4_kernel<Bf16Traits, false>), kDkv4Lds<128>);
    set(reinterpret_cast<const void*>(&fa_bwd_dkv4_kernel<F16Traits, true>), kDkv4Lds<128>);
    set(reinterpret_cast<const void*>(&fa_bwd_dkv4_kernel<F16Traits, false>), kDkv4Lds<128>);
    set(reinterpret_cast<const void*>(&fa_bwd_dkv4_kernel_d64<Bf16Traits, true>), kDkv4Lds<64>);
    set(reinterpret_cast<const void*>(&fa_bwd_dkv4_kernel_d64<Bf16Traits, false>), kDkv4Lds<64>);
    set(reinterpret_cast<const void*>(&fa_bwd_dkv4_kernel_d64<F16Traits, true>), kDkv4Lds<64>);
    set(reinterpret_cast<const void*>(&fa_bwd_dkv4_kernel_d64<F16Traits, false>), kDkv4Lds<64>);
    set(reinterpret_cast<const void*>(&fa_bwd_dkv4_kernel_d64k2<Bf16Traits, true>), kDkv4Lds<64>);
    set(reinterpret_cast<const void*>(&fa_bwd_dkv4_kernel_d64k2<Bf16Traits, false>), kDkv4Lds<64>);
    set(reinterpret_cast<const void*>(&fa_bwd_dkv4_kernel_d64k2<F16Traits, true>), kDkv4Lds<64>);
    set(reinterpret_cast<const void*>(&fa_bwd_dkv4_kernel_d64k2<F16Traits, false>), kDkv4Lds<64>);
    // the SPILL instances (5-matmul backward)
    set(reinterpret_cast<const void*>(&fa_bwd_dkv4_kernel<Bf16Traits, true, false, true>), kDkv4Lds<128>);
    set(reinterpret_cast<const void*>(&fa_bwd_dkv4_kernel<Bf16Traits, false, false, true>), kDkv4Lds<128>);
    set(reinterpret_cast<const void*>(&fa_bwd_dkv4_kernel<F16Traits, true, false, true>), kDkv4Lds<128>);
    set(reinterpret_cast<const void*>(&fa_bwd_dkv4_kernel<F16Traits, false, false, true>), kDkv4Lds<128>);
    set(reinterpret_cast<const void*>(&fa_bwd_dkv4_kernel_d64<Bf16Traits, true, true>), kDkv4Lds<64>);
    set(reinterpret_cast<const void*>(&fa_bwd_dkv4_kernel_d64<Bf16Traits, false, true>), kDkv4Lds<64>);
    set(reinterpret_cast<const void*>(&fa_bwd_dkv4_kernel_d64<F16Traits, true, true>), kDkv4Lds<64>);
    set(reinterpret_cast<const void*>(&fa_bwd_dkv4_kernel_d64<F16Traits, false, true>), kDkv4Lds<64>);
    return rc;
}

}  // namespace aule_hip
