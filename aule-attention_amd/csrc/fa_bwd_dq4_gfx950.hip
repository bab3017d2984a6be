// fa_bwd_dq4_gfx950.hip -- dQ of the FlashAttention-2 backward (16-bit I/O, D = 128 and, since round 4, D = 64), ONE WAVE PER SIMD.
//
// Replaces the dQ half of python/aule/triton_flash.py:242-350 / triton_flash_amd.py:247-351 (the reference's backward kernels)
// where it applies; everything else stays on fa_bwd_gfx950.hip's fa_bwd_dq_kernel (8 waves x 32 rows, two waves per SIMD).
// Why another kernel: DESIGN.md 7 item 3 -- a wave that owns 32 query rows needs 16 cycles of LDS issue per MFMA (K and V
// row-major + K transposed for 48 MFMAs per 64 keys); with 64 rows per wave every fragment read feeds two row blocks.  Here
//
//   * workgroup = 4 waves x 64 query rows = the predecessor's 256-row Q block (causal: the pair (i, n-1-i)); a wave keeps the
//     Q^T and dO^T fragments of its rows (128 accumulator registers) and dQ^T (128) for the whole part; scores, packed dS and the
//     K / V fragments in flight live in arch VGPRs -- all named literally by fa_bwd_dq4_asm.inc (generated: tools/gen_dq4.py);
//   * the wave walks the stream of 32-key KV blocks its Q block sees, three stages deep, ONE asm statement per iteration j:
//     barrier | S, dP of block j | arithmetic of block j-1 | dQ of block j-2 | LDS-DMA requests of block j+4;
//   * KV blocks arrive by LDS-DMA, one image per tensor (the dK/dV kernel's padded sub-tiles: ds_read_b128 and transpose reads
//     from the same image), in an 8-slot ring;
//   * a wave whose rows end below the workgroup's last KV blocks (causal) stops computing early and only keeps the barrier and
//     its share of the requests;
//   * delta = rowsum(O * dO), L' = LSE log2(e) and -delta are published for the dK/dV kernel exactly as the predecessor does.
//
// D = 64 (round 4): the same statement with 24 MFMAs per KV block (16 + 8) against the same 2 x 16 scores per lane; the block image is
// the dK/dV kernel's D = 64 one (a chunk permutation per 1 KB piece: tools/gen_bw4.py, chunk64), 128 accumulator registers, 50 arch
// VGPRs left to hipcc, a 68 KB ring.
//
// Covers bf16 / fp16, D = 128 / 64, causal (coff >= 0; round 5: with a sliding window: the stream starts at the first block the Q block's
// first row sees and every block runs a two-sided mask) and non-causal; deterministic (no atomics); the accumulation order
// over the keys is the predecessor's, so dQ comes out bit-identical to it.
#include <cstdlib>
#include <type_traits>

#include "fa_device.h"
#include "fa_kernels.h"
#include "fa_fwd_tile.h"

namespace aule_hip {
namespace {

#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
#ifdef DQ4_ASM_INC          // timing variants (tools/dq4_variants.sh)
#include DQ4_ASM_INC
#else
#include "fa_bwd_dq4_asm.inc"
#endif

struct Dq4Params {
    const void* q;
    const void* k;
    const void* v;
    const void* dout;
    const void* o;
    const float* lse;
    float* delta_out;    // [B, Hq, Sq]: delta, L' = LSE log2(e), - delta for the dK/dV kernel of the same call
    float* lse2_out;
    float* ndelta_out;
    void* dq;
    int B, Hq, Hkv, Sq, Sk;
    float c;       // scale * log2(e) (sign kept: no maximum is taken here)
    float scale;   // applied to dQ at the end
    int nblk;      // work items per (batch, q head): Q blocks, or pairs of them (causal)
    int coff;      // causal position offset (query i sits at position i + coff)
    int window;    // sliding window (round 5; causal only): key j visible to query i only if (i + coff) - j < window (0: off)
};

constexpr int kQBlock4 = 256;    // 4 waves x 64 query rows
constexpr int kKB4 = 32;         // keys per block of the stream
constexpr int kRingQ4 = 8;       // slots of the LDS ring (17 KB each): block j + 4 is requested while j - 2 .. j + 1 are read

template <int N>
__device__ __forceinline__ float dq4_acc_read() {
    float x = 0.f;
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("v_accvgpr_read_b32 %0, a%c1" : "=v"(x) : "n"(N));
#endif
    return x;
}

// accumulator block BASE + 16 d .. of the lane's query row -> the row's d = 32 d + 8 g + 4 hi .. + 3 (8-byte stores)
template <class T, int BASE, int NI, int I = 0>
__device__ __forceinline__ void dq4_store_rows(char* row, int hi, float sc) {
    if constexpr (I < NI) {
        constexpr int d = I / 4, g4 = I % 4, N = BASE + 16 * d + 4 * g4;
        u32x2_t u;
        u[0] = T::pack2(dq4_acc_read<N>() * sc, dq4_acc_read<N + 1>() * sc);
        u[1] = T::pack2(dq4_acc_read<N + 2>() * sc, dq4_acc_read<N + 3>() * sc);
        *reinterpret_cast<u32x2_t*>(row + (32 * d + 8 * g4 + 4 * hi) * 2) = u;
        dq4_store_rows<T, BASE, NI, I + 1>(row, hi, sc);
    }
}

__device__ __forceinline__ int dq4_rfl(int x) { return __builtin_amdgcn_readfirstlane(x); }

// the lane's half of rowsum(O * dO) over one row block: dwords OB .. OB + NI - 1 (O) and GB .. (dO) of the accumulator file, in the
// predecessor's order (k-slice by k-slice, dword by dword: the sums come out bit-identical to fa_bwd_dq_kernel's)
template <class T, int OB, int GB, int NI, int I = 0>
__device__ __forceinline__ float dq4_delta_part(float part) {
    if constexpr (I < NI) {
        unsigned ov = 0, gv = 0;
#if defined(__HIP_DEVICE_COMPILE__)
        // (`part` rides through the statement: without the tie hipcc issues all 64 reads first and keeps them live)
        asm volatile("v_accvgpr_read_b32 %0, a%c3\n\tv_accvgpr_read_b32 %1, a%c4" : "=v"(ov), "=v"(gv), "+v"(part) : "n"(OB + I), "n"(GB + I));
#endif
        part += T::lo(ov) * T::lo(gv) + T::hi(ov) * T::hi(gv);
        return dq4_delta_part<T, OB, GB, NI, I + 1>(part);
    } else {
        return part;
    }
}

template <class T, int D, bool CAUSAL>
__device__ __forceinline__ void dq4_body(const Dq4Params& p) {
    constexpr int RB = 2 * D, NF = D / 4;   // NF: dwords of a row block's Q^T (dO^T, O) fragments per lane
    using A = Dq4Asm<T, D>;
    using std::integral_constant;
    constexpr int SLOT = A::SLOT;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
    const int wave = dq4_rfl(tid >> 6);
#if defined(__HIP_DEVICE_COMPILE__)
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
#else
    const unsigned lds0 = 0;
#endif
    const int Sq = p.Sq, Sk = p.Sk, coff = p.coff;
    const float c = p.c;
    const WorkItem w = decode_work(blockIdx.x, p.B, p.Hq, p.Hkv, p.nblk, false);
    const int nqb = (Sq + kQBlock4 - 1) / kQBlock4;
    const size_t qbase = (size_t)(w.b * p.Hq + w.h) * Sq;
    const size_t kvhead = (size_t)(w.b * p.Hkv + w.hk) * Sk * RB;
    const int W = p.window;
    const __amdgpu_buffer_rsrc_t qrs = make_srd(reinterpret_cast<const char*>(p.q) + qbase * RB, (unsigned)Sq * RB);
    const __amdgpu_buffer_rsrc_t grs = make_srd(reinterpret_cast<const char*>(p.dout) + qbase * RB, (unsigned)Sq * RB);
    const __amdgpu_buffer_rsrc_t ors = make_srd(reinterpret_cast<const char*>(p.o) + qbase * RB, (unsigned)Sq * RB);

    // the image of a 32-row block (layout: tools/gen_bw4.py, Cfg): the lane constants of the dK/dV kernel (fa_bwd_dkv4_gfx950.hip)
    unsigned tr_off, a_sub, a_sub1 = 0, vost[2] = {0, 0}, wave_pb;
    if constexpr (D == 128) {
        // row group rg = row / 4 is a 1024-byte piece of eight [4 rows][16 d] sub-tiles at pbase(rg)
        auto pbase = [](int rg) { return 1024 * rg + (rg & 1) * 16 + ((rg >> 1) & 1) * 128 + (rg >> 2) * 256; };
        static_assert(D != 128 || (A::PB1 == 1040 && A::PB2 == 2048 + 128 && A::PB4 == 4096 + 256), "piece bases of the generator");
        tr_off = (unsigned)(hi * 1040 + ((lane >> 4) & 1) * 128 + (lane & 15) * 8);   // + the read's key-octet / d-block immediate
        a_sub = (unsigned)(pbase(l31 >> 2) + (l31 & 3) * 32 + hi * 16);               // key row l31, d = 16 ks + 8 hi ..: + 128 ks
        // per-lane source offsets of this wave's two pieces (row groups 2 w, 2 w + 1) of an image: LDS position = lane
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int rg = 2 * wave + h;
            vost[h] = (unsigned)((rg * 4 + ((lane >> 1) & 3)) * RB + ((lane >> 3) * 2 + (lane & 1)) * 16);
        }
        wave_pb = (unsigned)pbase(2 * wave);
    } else {
        // D = 64: piece p = rows 8 p .. 8 p + 7 at 1040 p; chunk (rgl, d, b, rr, h) at 16-byte position 32 d + chunk(rgl, b, rr, h)
        static_assert(D != 64 || (A::PB1 == 1040 && A::PB2 == 2080), "piece bases of the generator");
        auto chunk = [](int rgl, int b, int rr, int h) { return 16 * rgl + 8 * (rgl ^ b) + 2 * rr + (h ^ b); };
        tr_off = (unsigned)(16 * chunk(hi, (lane >> 4) & 1, (lane >> 2) & 3, (lane >> 1) & 1) + (lane & 1) * 8);   // + piece / d-block immediate
        a_sub = (unsigned)(1040 * (l31 >> 3) + 16 * chunk((l31 >> 2) & 1, 0, l31 & 3, hi));    // key row l31, k-slice 2 d + b: base of b, + 512 d
        a_sub1 = (unsigned)(1040 * (l31 >> 3) + 16 * chunk((l31 >> 2) & 1, 1, l31 & 3, hi));
        const int cd = lane >> 5, g5 = lane & 31, rgl = g5 >> 4, cb = rgl ^ ((g5 >> 3) & 1), rr = (g5 >> 1) & 3, ch = (g5 & 1) ^ cb;
        vost[0] = (unsigned)((wave * 8 + 4 * rgl + rr) * RB + (2 * cd + cb) * 32 + ch * 16);   // this wave's piece (piece w): lane l fills chunk l
        wave_pb = (unsigned)(1040 * wave);
    }
    auto slot_lds = [&](int x) __attribute__((always_inline)) { return lds0 + (unsigned)(x & (kRingQ4 - 1)) * SLOT; };

    const int nparts = (CAUSAL && (nqb - 1 - w.blk) != w.blk) ? 2 : 1;
    for (int part = 0; part < nparts; ++part) {
        const int qb = CAUSAL ? (part == 0 ? nqb - 1 - w.blk : w.blk) : w.blk;
        const int q0w = qb * kQBlock4 + wave * 64;
        // Sliding window: the workgroup's stream starts at the first 32-key block its FIRST row sees (block t0); everything below is
        // relative to that block -- the descriptors start there, block indices, key limits and mask thresholds count from it.
        const int t0 = (CAUSAL && W > 0) ? min(max(0, qb * kQBlock4 + coff - W + 1), Sk) / kKB4 : 0;   // (rows past Sk + W - 1 see no key: an empty range at the end of K)
        const int k00 = t0 * kKB4;
        const __amdgpu_buffer_rsrc_t krs = make_srd(reinterpret_cast<const char*>(p.k) + kvhead + (size_t)k00 * RB, (unsigned)dq4_rfl((Sk - k00) * RB));
        const __amdgpu_buffer_rsrc_t vrs = make_srd(reinterpret_cast<const char*>(p.v) + kvhead + (size_t)k00 * RB, (unsigned)dq4_rfl((Sk - k00) * RB));
        const int kv_hi = (CAUSAL ? min(Sk, qb * kQBlock4 + kQBlock4 + coff) : Sk) - k00;     // keys the workgroup's rows can see (from k00)
        const int n = max(0, (kv_hi + kKB4 - 1) / kKB4);                                         // blocks of the workgroup's stream
        const int kv_hi_w = (CAUSAL ? min(Sk, q0w + 64 + coff) : Sk) - k00;                    // ... this wave's rows
        const int n_w = dq4_rfl(min(n, max(0, (kv_hi_w + kKB4 - 1) / kKB4)));

        // ---- Q^T, dO^T fragments into the accumulator file; delta, L' of the lane's two rows (published for the dK/dV kernel)
        A::load_frags(qrs, grs, ors, (unsigned)((q0w + l31) * RB + hi * 16), (unsigned)((q0w + 32 + l31) * RB + hi * 16));
        int lim4[2], wd4[2] = {0, 0};
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
            const int qrow = q0w + 32 * rb + l31;
            const int qr = qrow < Sq ? qrow : Sq - 1;
            const float part_sum = rb == 0 ? dq4_delta_part<T, 0, A::DF, NF>(0.f) : dq4_delta_part<T, NF, A::DF + NF, NF>(0.f);
            const float delta = part_sum + xhalf(part_sum);
            const float nlse2 = -p.lse[qbase + qr] * kLog2e;
            if (hi == 0 && qrow < Sq) {
                p.delta_out[qbase + qrow] = delta;
                p.lse2_out[qbase + qrow] = -nlse2;
                p.ndelta_out[qbase + qrow] = -delta;
            }
            if (rb == 0) A::set_scal0(-nlse2, delta); else A::set_scal1(-nlse2, delta);
            // last key visible to the lane's row (minus 4 hi: a score register r holds key crow(r) + 4 hi of its block), from k00
            const int last = (CAUSAL ? min(Sk - 1, qrow + coff) : Sk - 1) - k00;
            lim4[rb] = last - 4 * hi;
            if (CAUSAL && W > 0) {   // window: lim4 = the FIRST visible key (minus 4 hi), wd4 = the number of visible keys (AR = 3 statements)
                const int first = max(0, qrow + coff - W + 1) - k00;
                lim4[rb] = first - 4 * hi;
                wd4[rb] = max(0, last - first + 1);
            }
        }
        A::zero_acc();
        // blocks b >= mask_lo need the mask for some row of the wave (causal diagonal); so does a ragged last block of K
        const int mask_lo = dq4_rfl(CAUSAL ? max(0, q0w + coff + 1 - k00) / kKB4 : 0x7fffffff);
        const int ragged_blk = dq4_rfl((Sk % kKB4) != 0 ? Sk / kKB4 - t0 : -1);
        const bool win = CAUSAL && W > 0;
        // window: blocks whose first key lies at or above the window start of the wave's LAST row are free of the lower bound -- between them
        // and the diagonal the plain arithmetic runs (W = 256: 6 of a wave's 11 blocks; W = 1024: 30 of 35)
        const int win_lo = dq4_rfl(win ? max(0, (q0w + 63 + coff - W + 1 - k00 + kKB4 - 1) / kKB4) : 0);

        // ---- stream start: blocks 0 .. 3 requested (a block behind the stream reads zeros: the scalar offset is range-checked)
#pragma unroll
        for (int x = 0; x < 4; ++x)
            A::dma_block(slot_lds(x) + wave_pb, krs, vrs, (unsigned)(x * kKB4 * RB), vost[0], vost[1]);

        // iteration j: S / dP of block j, arithmetic of block j - 1, dQ of block j - 2 -- whatever of them exists for this wave
        auto iteration = [&](auto par_tag, int j) __attribute__((always_inline)) {
            constexpr int PAR = decltype(par_tag)::value;
            const unsigned ra = slot_lds(j) + a_sub, ra2 = slot_lds(j + 1) + a_sub, trb = slot_lds(j - 2) + tr_off;
            const unsigned rab = slot_lds(j) + a_sub1, ra2b = slot_lds(j + 1) + a_sub1;   // (D = 64: the bases of the odd k-slices)
            const unsigned dlds = slot_lds(j + 4) + wave_pb, dso = (unsigned)((j + 4) * kKB4 * RB);
            const int k0 = (j - 1) * kKB4;
#define DQ4_IT(QK, NXT, AR, DQ, PRE) A::template iter<PAR, QK, NXT, AR, DQ, PRE>(c, ra, rab, ra2, ra2b, trb, lim4[0], lim4[1], k0, dlds, krs, vrs, dso, vost[0], vost[1], wd4[0], wd4[1])
            const bool plain = (j - 1) >= win_lo && (j - 1) < mask_lo && (j - 1) != ragged_blk;
            if (j + 1 < n_w) {            // S / dP of block j, and of block j + 1 next time
                if (j >= 2) { if (plain) DQ4_IT(1, 1, 1, 1, 1); else if (win) DQ4_IT(1, 1, 3, 1, 1); else DQ4_IT(1, 1, 2, 1, 1); }
                else { if (plain) DQ4_IT(1, 1, 1, 0, 1); else if (win) DQ4_IT(1, 1, 3, 0, 1); else DQ4_IT(1, 1, 2, 0, 1); }
            } else if (j < n_w) {         // the wave's last S / dP
                if (j >= 2) { if (plain) DQ4_IT(1, 0, 1, 1, 1); else if (win) DQ4_IT(1, 0, 3, 1, 1); else DQ4_IT(1, 0, 2, 1, 1); }
                else { if (plain) DQ4_IT(1, 0, 1, 0, 1); else if (win) DQ4_IT(1, 0, 3, 0, 1); else DQ4_IT(1, 0, 2, 0, 1); }
            } else if (j - 1 < n_w) {     // tail: arithmetic of the last block (+ dQ of the one before)
                if (j >= 2) { if (plain) DQ4_IT(0, 0, 1, 1, 0); else if (win) DQ4_IT(0, 0, 3, 1, 0); else DQ4_IT(0, 0, 2, 1, 0); }
                else { if (plain) DQ4_IT(0, 0, 1, 0, 0); else if (win) DQ4_IT(0, 0, 3, 0, 0); else DQ4_IT(0, 0, 2, 0, 0); }
            } else if (j - 2 < n_w && j >= 2) {
                DQ4_IT(0, 0, 0, 1, 0);   // dQ of the last block
            } else {
                DQ4_IT(0, 0, 0, 0, 0);   // idle: barrier + this wave's share of the requests
            }
        };
        {   // iteration 0 requests its own first fragments
            const unsigned ra = slot_lds(0) + a_sub, ra2 = slot_lds(1) + a_sub, rab = slot_lds(0) + a_sub1, ra2b = slot_lds(1) + a_sub1;
            const unsigned dlds = slot_lds(4) + wave_pb, dso = (unsigned)(4 * kKB4 * RB);
            if (n_w > 1) A::template iter<0, 1, 1, 0, 0, 0>(c, ra, rab, ra2, ra2b, 0, 0, 0, 0, dlds, krs, vrs, dso, vost[0], vost[1]);
            else if (n_w > 0) A::template iter<0, 1, 0, 0, 0, 0>(c, ra, rab, ra2, ra2b, 0, 0, 0, 0, dlds, krs, vrs, dso, vost[0], vost[1]);
            else A::template iter<0, 0, 0, 0, 0, 0>(c, ra, rab, ra2, ra2b, 0, 0, 0, 0, dlds, krs, vrs, dso, vost[0], vost[1]);
        }
#undef DQ4_IT
        // The steady range -- S / dP of block j with block j + 1 to follow, plain arithmetic of block j - 1, dQ of block j - 2 -- runs
        // without the variant decision: one wave per SIMD pays ~4.6 cycles for every scalar instruction between two statements
        // while the matrix pipe drains (profiles/r3b_bwd_dq4.txt).  Everything else goes through iteration().
        auto steady = [&](auto par_tag, int j) __attribute__((always_inline)) {
            constexpr int PAR = decltype(par_tag)::value;
            A::template iter<PAR, 1, 1, 1, 1, 1>(c, slot_lds(j) + a_sub, slot_lds(j) + a_sub1, slot_lds(j + 1) + a_sub, slot_lds(j + 1) + a_sub1,
                                                slot_lds(j - 2) + tr_off, 0, 0, 0,
                                                slot_lds(j + 4) + wave_pb, krs, vrs, (unsigned)((j + 4) * kKB4 * RB), vost[0], vost[1]);
        };
        int steady_end = min(n_w - 1, mask_lo == 0x7fffffff ? mask_lo : mask_lo + 1);   // j + 1 < n_w and block j - 1 below the diagonal
        if (ragged_blk >= 0) steady_end = min(steady_end, ragged_blk + 1);               // ... and not the ragged last block of K
        if (win) steady_end = 0;                                                          // (window: no unmasked steady range)
        int j = 1;
        iteration(integral_constant<int, 1>{}, j++);
        for (; j + 1 < steady_end; j += 2) {      // (j is even here)
            steady(integral_constant<int, 0>{}, j);
            steady(integral_constant<int, 1>{}, j + 1);
        }
        for (; j <= n + 1; ++j) {
            if (j & 1) iteration(integral_constant<int, 1>{}, j);
            else iteration(integral_constant<int, 0>{}, j);
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");   // the requests behind the stream too; the ring is free

        // ---- dQ (scaled) of the lane's two rows
        asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");   // the last MFMAs -> v_accvgpr_read
        {
            int lane_o = lane;
            asm volatile("" : "+v"(lane_o));
            const int r0 = q0w + (lane_o & 31);
            if (r0 < Sq) dq4_store_rows<T, 0, D / 8>(reinterpret_cast<char*>(p.dq) + (qbase + r0) * RB, lane_o >> 5, p.scale);
            if (r0 + 32 < Sq) dq4_store_rows<T, D / 2, D / 8>(reinterpret_cast<char*>(p.dq) + (qbase + r0 + 32) * RB, lane_o >> 5, p.scale);
        }
        __syncthreads();
    }
}

template <class T, bool CAUSAL>
__global__ void __launch_bounds__(256, 1) __attribute__((amdgpu_num_vgpr(34))) fa_bwd_dq4_kernel(const Dq4Params p) {
    static_assert(Dq4Asm<T, 128>::NV == 34, "amdgpu_num_vgpr must be the generator's NV");
    dq4_body<T, 128, CAUSAL>(p);
}

// D = 64: 50 arch VGPRs for hipcc (the attribute takes a literal, hence a kernel of its own)
template <class T, bool CAUSAL>
__global__ void __launch_bounds__(256, 1) __attribute__((amdgpu_num_vgpr(50))) fa_bwd_dq4_kernel_d64(const Dq4Params p) {
    static_assert(Dq4Asm<T, 64>::NV == 50, "amdgpu_num_vgpr must be the generator's NV");
    dq4_body<T, 64, CAUSAL>(p);
}

#pragma clang diagnostic pop

template <int D>
constexpr int kDq4Lds = kRingQ4 * Dq4Asm<Bf16Traits, D>::SLOT;

template <class T, int D>
int launch_dq4(const BwdArgs& a, float* lse2_out, float* ndelta_out, hipStream_t stream) {
    Dq4Params p;
    p.q = a.q; p.k = a.k; p.v = a.v; p.dout = a.dout; p.o = a.o; p.lse = a.lse;
    p.delta_out = a.delta; p.lse2_out = lse2_out; p.ndelta_out = ndelta_out;
    p.dq = a.dq;
    p.B = a.B; p.Hq = a.Hq; p.Hkv = a.Hkv; p.Sq = a.Sq; p.Sk = a.Sk;
    p.c = a.scale * kLog2e;
    p.scale = a.scale;
    p.coff = a.causal ? a.coff : 0;
    p.window = (a.causal && a.window > 0) ? a.window : 0;
    const int nqb = (a.Sq + kQBlock4 - 1) / kQBlock4;
    p.nblk = a.causal ? (nqb + 1) / 2 : nqb;
    const dim3 grid((unsigned)(p.nblk * a.B * a.Hq)), block(256);
    if constexpr (D == 64) {
        if (a.causal)
            hipLaunchKernelGGL((fa_bwd_dq4_kernel_d64<T, true>), grid, block, kDq4Lds<64>, stream, p);
        else
            hipLaunchKernelGGL((fa_bwd_dq4_kernel_d64<T, false>), grid, block, kDq4Lds<64>, stream, p);
    } else {
        if (a.causal)
            hipLaunchKernelGGL((fa_bwd_dq4_kernel<T, true>), grid, block, kDq4Lds<128>, stream, p);
        else
            hipLaunchKernelGGL((fa_bwd_dq4_kernel<T, false>), grid, block, kDq4Lds<128>, stream, p);
    }
    return (int)hipGetLastError();
}

}  // namespace

// Shapes the one-wave-per-SIMD dQ kernel can take: 16-bit, D = 128 or 64, no window, causal offset >= 0, offsets inside 2 GB
// descriptors.  AULE_HIP_BWD_DQ=new takes it wherever it can run, =old never (A/B, tests); default: the dispatcher's grid rule
// (fa_bwd_gfx950.hip).
int bwd_dq4_mode() {
    static const int mode = [] {
        const char* e = std::getenv("AULE_HIP_BWD_DQ");
        return e == nullptr ? 0 : (e[0] == 'o' ? 1 : (e[0] == 'n' ? 2 : 0));
    }();
    return mode;
}

bool bwd_dq4_applicable(const BwdArgs& a) {
    if (bwd_dq4_mode() == 1) return false;
    if (a.dtype != kBF16 && a.dtype != kF16) return false;
    if (a.D != 128 && a.D != 64) return false;
    if (a.window > 0 && !a.causal) return false;      // (round 5: causal sliding windows run here; a window without the causal rule stays on the predecessor)
    if (a.causal && a.coff < 0) return false;
    if (a.Hkv <= 0 || a.Hq % a.Hkv != 0) return false;
    if ((long long)a.Sq * a.D * 2 >= (1LL << 31) || ((long long)a.Sk + 6 * kKB4) * a.D * 2 >= (1LL << 31)) return false;
    return true;
}

int launch_bwd_dq4(const BwdArgs& a, float* lse2_out, float* ndelta_out, hipStream_t stream) {
    if (a.D == 128) {
        if (a.dtype == kBF16) return launch_dq4<Bf16Traits, 128>(a, lse2_out, ndelta_out, stream);
        if (a.dtype == kF16) return launch_dq4<F16Traits, 128>(a, lse2_out, ndelta_out, stream);
    } else if (a.D == 64) {
        if (a.dtype == kBF16) return launch_dq4<Bf16Traits, 64>(a, lse2_out, ndelta_out, stream);
        if (a.dtype == kF16) return launch_dq4<F16Traits, 64>(a, lse2_out, ndelta_out, stream);
    }
    return -1;
}

int configure_bwd_dq4() {
    int rc = 0;
    auto set = [&](const void* f, int lds) { rc |= (int)hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, lds); };
    set(reinterpret_cast<const void*>(&fa_bwd_dq4_kernel<Bf16Traits, true>), kDq4Lds<128>);
    set(reinterpret_cast<const void*>(&fa_bwd_dq4_kernel<Bf16Traits, false>), kDq4Lds<128>);
    set(reinterpret_cast<const void*>(&fa_bwd_dq4_kernel<F16Traits, true>), kDq4Lds<128>);
    set(reinterpret_cast<const void*>(&fa_bwd_dq4_kernel<F16Traits, false>), kDq4Lds<128>);
    set(reinterpret_cast<const void*>(&fa_bwd_dq4_kernel_d64<Bf16Traits, true>), kDq4Lds<64>);
    set(reinterpret_cast<const void*>(&fa_bwd_dq4_kernel_d64<Bf16Traits, false>), kDq4Lds<64>);
    set(reinterpret_cast<const void*>(&fa_bwd_dq4_kernel_d64<F16Traits, true>), kDq4Lds<64>);
    set(reinterpret_cast<const void*>(&fa_bwd_dq4_kernel_d64<F16Traits, false>), kDq4Lds<64>);
    return rc;
}

}  // namespace aule_hip
