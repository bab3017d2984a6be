// fa_bwd_dqs_gfx950.hip -- the dQ half of the FIVE-matmul backward (round 5; 16-bit I/O, D = 128 / 64), and its delta pre-pass.
//
// Replaces the dQ half of python/aule/triton_flash_amd.py:247-351 / triton_flash.py:242-350 (the reference's backward kernels).
// Rounds 1-4 ran two deterministic kernels that each recomputed S = Q K^T and dP = dO V^T (7 tile matmuls for 5: 1.47 x the
// algorithmic MFMAs, and on MI355X the backward runs at the socket's power limit, where a launch's time is its energy).  Now:
//
//   1. fa_bwd_delta16_kernel (here, HBM-bound): delta = rowsum(O * dO), - delta, L' = LSE log2(e) -> workspace;
//   2. fa_bwd_dkv4_kernel<.., SPILL> (fa_bwd_dkv4_gfx950.hip): S, dP, dV, dK as before, and the packed 16-bit dS of every
//      (32 keys x 32 rows) tile -- the very registers it feeds to its dK MFMAs -- goes to the workspace (2 KB per tile, two stores);
//   3. fa_bwd_dqs_kernel (here): dQ = scale * dS K -- ONE matmul, no exponentials: streams the dS tiles of its 256 query rows back
//      (LDS-DMA, then ds_read_b64_tr_b16: the tile was written key-major, the MFMA wants it query-major) next to the K blocks.
//
// 5 tile matmuls, no atomics, bit-deterministic; the price is one 16-bit round trip of dS through HBM / the Infinity Cache
// (C3: 2 x 0.57 GB per backward, C2: 2 x 2.2 GB) and O(Sq Sk) workspace, which the dispatcher (fa_bwd_gfx950.hip) bounds by
// running the batch in chunks.  This kernel is bound by that stream (2 KB of dS per 8 MFMAs: 16 TB/s to feed the matrix pipes; measured:
// 4.2 TB/s from HBM, the same with the MFMAs removed -- profiles/r5_bwd_spill.txt), not by MFMA issue, so it is plain HIP: compiler-scheduled
// MFMAs and LDS reads around hand-issued LDS-DMA requests with a counted vmcnt; a ring of 3 block slots, blocks requested 2 iterations
// ahead, two workgroups per CU (deeper rings and one workgroup per CU measured the same).
// MEASURED VERDICT: a loss against the recompute pair wherever the dS leaves the 256 MB Infinity Cache (C2, C3, D = 64
// training), +14 .. 27 % where it stays inside -- the default dispatch takes this mode for those problems only.
//
// Workspace layout: fa_kernels.h, DsLayout.  Work decomposition = the recompute kernel's (fa_bwd_dq4_gfx950.hip): a workgroup of
// 4 waves x 64 query rows owns a 256-row Q block (causal: the pair (i, n-1-i); single blocks when every block can have a workgroup
// slot of its own) and walks the 32-key blocks its rows see.
#include <cstdlib>

#include "fa_device.h"
#include "fa_kernels.h"
#include "fa_fwd_tile.h"

namespace aule_hip {
namespace {

// ---------------------------------------------------------------------------------------------------------------- delta pass
struct Delta16Params {
    const void* o;
    const void* dout;
    const float* lse;
    float* delta;    // [B, Hq, Sq] each
    float* lse2;
    float* ndelta;
    long long rows;
};

// LPR = D / 8 lanes per row (one 16-byte chunk of O and of dO each), four rows per thread in flight
template <class T, int D>
__global__ void __launch_bounds__(256) fa_bwd_delta16_kernel(const Delta16Params p) {
    constexpr int LPR = D / 8, RPB = 256 / LPR, UN = 4;
    const int tid = threadIdx.x, c = tid % LPR;
    const long long row0 = (long long)blockIdx.x * (RPB * UN) + tid / LPR;
    u32x4_t ov[UN], gv[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
        const long long row = row0 + u * RPB;
        const long long r = row < p.rows ? row : p.rows - 1;
        ov[u] = *reinterpret_cast<const u32x4_t*>(reinterpret_cast<const char*>(p.o) + (r * D + c * 8) * 2);
        gv[u] = *reinterpret_cast<const u32x4_t*>(reinterpret_cast<const char*>(p.dout) + (r * D + c * 8) * 2);
    }
#pragma unroll
    for (int u = 0; u < UN; ++u) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) s += T::lo(ov[u][i]) * T::lo(gv[u][i]) + T::hi(ov[u][i]) * T::hi(gv[u][i]);
#pragma unroll
        for (int m = LPR / 2; m >= 1; m >>= 1) s += __shfl_xor(s, m, 64);
        const long long row = row0 + u * RPB;
        if (c == 0 && row < p.rows) {
            p.delta[row] = s;
            p.ndelta[row] = -s;
            p.lse2[row] = p.lse[row] * kLog2e;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------- dQ = dS K
struct DqsParams {
    const void* k;
    const char* ds;
    void* dq;
    int B, Hq, Hkv, Sq, Sk;
    float scale;   // applied to dQ at the end
    int nblk;      // work items per (batch, q head): Q blocks, or pairs of them (causal)
    int coff;
    int nq32, nkb32p;
    int rev;       // walk the grid backwards (what the dK/dV kernel wrote last is read first: Infinity Cache)
    int pair;      // causal: 1 = a work item is the pair of Q blocks (i, n-1-i) (equal work per item), 0 = one block per item (half-empty grids)
};

constexpr int kDqsQBlock = 256;   // 4 waves x 64 query rows
constexpr int kDqsKB = 32;        // keys per block of the stream
#ifndef DQS_SLOTS
#define DQS_SLOTS 3
#endif
#ifndef DQS_AHEAD
#define DQS_AHEAD 2
#endif
#ifndef DQS_OCC
#define DQS_OCC 2          // workgroups per CU: two 3-slot rings (150 KB of LDS), two waves per SIMD -- one workgroup's DMA issue and
#endif                     // transpose reads run under the other's MFMAs (a single workgroup serialises them behind its barrier)
constexpr int kDqsSlots = DQS_SLOTS;      // LDS ring
constexpr int kDqsAhead = DQS_AHEAD;      // blocks requested ahead (< kDqsSlots: the slot of block j + AHEAD is block j - 1's)
static_assert(kDqsAhead < kDqsSlots, "the slot of block j + AHEAD must be block j - 1's or older");
// timing experiments only (results are garbage): DQS_X_NOMFMA no tr-reads / MFMAs (the stream alone), DQS_X_SEQ a layout in which a
// wave's units are consecutive in memory, DQS_X_NODS no dS requests (the K side alone)

template <int D>
struct DqsCfg {
    static constexpr int RB = 2 * D;
    static constexpr int IMGK = D == 128 ? 8704 : 4352;   // the dK/dV kernel's image of a 32-row block (tools/gen_bw4.py, Cfg)
    static constexpr int NPK = D == 128 ? 2 : 1;          // K pieces (1 KB) per wave and block
    static constexpr int NP = NPK + 4;                    // + the wave's own two dS units
    static constexpr int SLOT = IMGK + 4 * 2 * 2048;
    static constexpr int LDS = kDqsSlots * SLOT;
    static constexpr int DB = D / 32;
};

__device__ __forceinline__ int dqs_rfl(int x) { return __builtin_amdgcn_readfirstlane(x); }

// one LDS-DMA piece: 64 lanes x 16 bytes from (descriptor, lane offset, scalar offset) to LDS address lds + 16 lane
__device__ __forceinline__ void dqs_dma(unsigned lds, unsigned voff, __amdgpu_buffer_rsrc_t srd, unsigned soff) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lds), "v"(voff), "s"(srd), "s"(soff) : "memory", "m0");
#endif
}

template <class T, int D, bool CAUSAL>
__global__ void __launch_bounds__(256, DQS_OCC) fa_bwd_dqs_kernel(const DqsParams p) {
    using C = DqsCfg<D>;
    using v8 = typename T::v8;
    constexpr int RB = C::RB, DB = C::DB;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
    const int wave = dqs_rfl(tid >> 6);
#if defined(__HIP_DEVICE_COMPILE__)
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
#else
    const unsigned lds0 = 0;
#endif
    const int Sq = p.Sq, Sk = p.Sk, coff = p.coff;
    const int bid = p.rev ? (int)(gridDim.x - 1 - blockIdx.x) : (int)blockIdx.x;
    const WorkItem w = decode_work(bid, p.B, p.Hq, p.Hkv, p.nblk, false);
    const int g = p.Hq / p.Hkv, hh = w.h - w.hk * g;
    const int nqb = (Sq + kDqsQBlock - 1) / kDqsQBlock;
    const int nkb32 = (Sk + kDqsKB - 1) / kDqsKB;
    const size_t qbase = (size_t)(w.b * p.Hq + w.h) * Sq;
    const size_t grp = (size_t)(w.b * p.Hkv + w.hk);
    const long long xs = (long long)g * p.nq32;
    const __amdgpu_buffer_rsrc_t krs = make_srd(reinterpret_cast<const char*>(p.k) + grp * Sk * RB, (unsigned)Sk * RB);
    const __amdgpu_buffer_rsrc_t srs = make_srd(p.ds + grp * (size_t)p.nkb32p * xs * 2048, (unsigned)((long long)p.nkb32p * xs * 2048));

    // ---- lane constants.  K: the image, the DMA source offsets and the transpose-read base of fa_bwd_dq4_gfx950.hip
    unsigned ktr, kvo[2] = {0, 0}, kpb;
    if constexpr (D == 128) {
        auto pbase = [](int rg) { return 1024 * rg + (rg & 1) * 16 + ((rg >> 1) & 1) * 128 + (rg >> 2) * 256; };
        ktr = (unsigned)(hi * 1040 + ((lane >> 4) & 1) * 128 + (lane & 15) * 8);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int rg = 2 * wave + h;
            kvo[h] = (unsigned)((rg * 4 + ((lane >> 1) & 3)) * RB + ((lane >> 3) * 2 + (lane & 1)) * 16);
        }
        kpb = (unsigned)pbase(2 * wave);
    } else {
        auto chunk = [](int rgl, int b, int rr, int h) { return 16 * rgl + 8 * (rgl ^ b) + 2 * rr + (h ^ b); };
        ktr = (unsigned)(16 * chunk(hi, (lane >> 4) & 1, (lane >> 2) & 3, (lane >> 1) & 1) + (lane & 1) * 8);
        const int cd = lane >> 5, g5 = lane & 31, rgl = g5 >> 4, cb = rgl ^ ((g5 >> 3) & 1), rr = (g5 >> 1) & 3, ch = (g5 & 1) ^ cb;
        kvo[0] = (unsigned)((wave * 8 + 4 * rgl + rr) * RB + (2 * cd + cb) * 32 + ch * 16);
        kpb = (unsigned)(1040 * wave);
    }
    // dS: a unit arrives as two 1 KB pieces; LDS chunk 64 p + l of the unit's image holds (key n = 16 p + (l >> 2), query step
    // kk = (l >> 1) & 1, hi = l & 1), i.e. the image is [key][kk][hi][16 bytes] -- a 32-lane pass of the transpose reads below then
    // covers 256 contiguous bytes.  Source (DsLayout): kk * 1024 + (n + 32 hi) * 16.
#ifndef DQS_DMA_QUAD
#define DQS_DMA_QUAD 1
#endif
#if DQS_DMA_QUAD
    // (round 5, session 2) the image of a unit is made of 256-byte windows of four keys, window w = keys 4 w .. 4 w + 3 as [c = 2 kk + hi][key & 3]
    // 16-byte chunks: four consecutive lanes of a piece then fetch 64 CONTIGUOUS source bytes (the first build's [key][kk][hi] order
    // sent every lane of a quad to another 128-byte line: 3.3 TB/s), and a 32-lane pass of the transpose reads still covers one
    // whole window.  LDS chunk 64 p + l: window 4 p + (l >> 4), c = (l >> 2) & 3, key & 3 = l & 3.
    const unsigned svo = (unsigned)(((lane >> 3) & 1) * 1024 + ((lane >> 2) & 1) * 512 + (4 * (lane >> 4) + (lane & 3)) * 16);   // piece 0; piece 1: + 256
    // transpose read (kk2 = 16-key step, e): lane (hi, qhalf = bit 4, i = lane & 15) addresses key 16 kk2 + 8 e + 4 hi + (i >> 2)
    // (window 4 kk2 + 2 e + hi, key & 3 = i >> 2), query rows 16 qhalf + 4 (i & 3) .. + 3 = (kk = qhalf, hi' = i & 1, half = (i >> 1) & 1),
    // and receives query 16 qhalf + i, 4 keys
    const unsigned str = (unsigned)(256 * hi + 64 * (2 * ((lane >> 4) & 1) + (lane & 1)) + 16 * ((lane & 15) >> 2) + 8 * ((lane >> 1) & 1));
    constexpr int kStrKK = 1024, kStrE = 512;
#else
    const unsigned svo = (unsigned)(((lane >> 1) & 1) * 1024 + (lane & 1) * 512 + (lane >> 2) * 16);   // piece 0; piece 1: + 256
    // transpose read (kk2 = 16-key step, e): lane (hi, qhalf = bit 4, i = lane & 15) addresses key 16 kk2 + 8 e + 4 hi + (i >> 2),
    // query rows 16 qhalf + 4 (i & 3) .. + 3 = (kk = qhalf, hi' = i & 1, half = (i >> 1) & 1) and receives query 16 qhalf + i, 4 keys
    const unsigned str = (unsigned)(256 * hi + 64 * ((lane & 15) >> 2) + 32 * ((lane >> 4) & 1) + 16 * (lane & 1) + 8 * ((lane >> 1) & 1));
    constexpr int kStrKK = 1024, kStrE = 512;
#endif
    const unsigned sun = (unsigned)(C::IMGK + wave * 4096);   // this wave's two units inside a slot

    const bool paired = CAUSAL && p.pair != 0;
    const int nparts = (paired && (nqb - 1 - w.blk) != w.blk) ? 2 : 1;
    for (int part = 0; part < nparts; ++part) {
        const int qb = paired ? (part == 0 ? nqb - 1 - w.blk : w.blk) : w.blk;
        const int q0w = qb * kDqsQBlock + wave * 64;
        // blocks of the workgroup's stream / of each of this wave's two row blocks (every unit they name was written: a unit exists
        // wherever its 128-key block is seen by ANY row of the 32-row query block)
        const int n = CAUSAL ? min(nkb32, (qb * kDqsQBlock + kDqsQBlock - 1 + coff) / kDqsKB + 1) : nkb32;
        int n_rb[2];
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
            const int qlo = q0w + 32 * rb;
            n_rb[rb] = dqs_rfl(qlo < Sq ? (CAUSAL ? min(nkb32, (qlo + 31 + coff) / kDqsKB + 1) : nkb32) : 0);
        }
        // unit of (row block rb, key block j): column j, position g qb32 + hh (block-major, head-minor: the dK/dV kernel's stream order
        // since round 6, fa_kernels.h DsLayout)
        const int u0 = (q0w / 32) * g + hh;
        auto ds_off = [&](int j, int rb) __attribute__((always_inline)) {
#ifdef DQS_X_SEQ
            return (unsigned)(((long long)(u0 + rb * g) * p.nkb32p + j) << 11);
#else
            return (unsigned)(((long long)j * xs + u0 + rb * g) << 11);
#endif
        };
        auto issue = [&](int j, int slot) __attribute__((always_inline)) {
            const unsigned sl = lds0 + (unsigned)slot * C::SLOT;
            const unsigned ko = (unsigned)(j * kDqsKB * RB);
            dqs_dma(dqs_rfl((int)(sl + kpb)), kvo[0], krs, ko);
            if constexpr (D == 128) dqs_dma(dqs_rfl((int)(sl + kpb + 1040)), kvo[1], krs, ko);
#pragma unroll
            for (int rb = 0; rb < 2; ++rb) {
                const unsigned so = ds_off(j, rb);
#ifdef DQS_X_NODS
                dqs_dma(dqs_rfl((int)(sl + sun + rb * 2048)), svo, krs, ko);
                dqs_dma(dqs_rfl((int)(sl + sun + rb * 2048 + 1024)), svo, krs, ko + 0 * so);
#else
                dqs_dma(dqs_rfl((int)(sl + sun + rb * 2048)), svo, srs, so);
                dqs_dma(dqs_rfl((int)(sl + sun + rb * 2048 + 1024)), svo, srs, so + 256u);
#endif
            }
        };

        f32x16_t acc[2][DB];
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int d = 0; d < DB; ++d)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[rb][d][r] = 0.f;

#pragma unroll
        for (int x = 0; x < kDqsAhead; ++x) issue(x, x);
        int slot = 0, dslot = kDqsAhead;
        for (int j = 0; j < n; ++j) {
            // block j has landed for this wave (all but the newest AHEAD - 1 blocks' pieces are complete) -> for everybody; every
            // LDS read of block j - 1 has returned: its slot is free for block j + AHEAD
            asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"((kDqsAhead - 1) * C::NP) : "memory");
            issue(j + kDqsAhead, dslot);
            const char* sl = smem + slot * C::SLOT;
#ifndef DQS_X_NOMFMA
            v8 kt[2][DB];
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int d = 0; d < DB; ++d) {
                    int o0, o1;
                    if constexpr (D == 128) {
                        auto pbase = [](int rg) { return 1024 * rg + (rg & 1) * 16 + ((rg >> 1) & 1) * 128 + (rg >> 2) * 256; };
                        o0 = pbase(2 * (2 * kk)) + 256 * d;
                        o1 = pbase(2 * (2 * kk + 1)) + 256 * d;
                    } else {
                        o0 = 1040 * (2 * kk) + 512 * d;
                        o1 = 1040 * (2 * kk + 1) + 512 * d;
                    }
                    kt[kk][d] = as_v8<T>(lds_tr16(sl + ktr + o0), lds_tr16(sl + ktr + o1));
                }
#pragma unroll
            for (int rb = 0; rb < 2; ++rb) {
                // a row block that does not see block j (causal: the workgroup's later rows do) multiplies zeros: its unit may not
                // even have been written.  A mask, not a branch: a branch makes hipcc copy the 64 accumulator registers around it.
                const unsigned live = j < n_rb[rb] ? 0xffffffffu : 0u;
                const char* un = sl + sun + rb * 2048 + str;
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    u32x2_t a = __builtin_bit_cast(u32x2_t, lds_tr16(un + kStrKK * kk));
                    u32x2_t b = __builtin_bit_cast(u32x2_t, lds_tr16(un + kStrKK * kk + kStrE));
                    const u32x4_t m = {a[0] & live, a[1] & live, b[0] & live, b[1] & live};
                    const v8 dsv = as_v8<T>(m);
#pragma unroll
                    for (int d = 0; d < DB; ++d) acc[rb][d] = T::mfma(kt[kk][d], dsv, acc[rb][d]);
                }
            }
#else
            (void)sl; (void)ktr; (void)str;
#endif
            slot = slot + 1 == kDqsSlots ? 0 : slot + 1;
            dslot = dslot + 1 == kDqsSlots ? 0 : dslot + 1;
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");   // the requests behind the stream too; the ring is free

        // ---- dQ (scaled) of the lane's two rows: acc[rb][d][r] = dQ^T[32 d + crow(r, hi)][row l31]
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
            const int r0 = q0w + 32 * rb + l31;
            if (r0 < Sq) {
                char* row = reinterpret_cast<char*>(p.dq) + (qbase + r0) * RB;
#pragma unroll
                for (int d = 0; d < DB; ++d)
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        u32x2_t u;
                        u[0] = T::pack2(acc[rb][d][4 * g4] * p.scale, acc[rb][d][4 * g4 + 1] * p.scale);
                        u[1] = T::pack2(acc[rb][d][4 * g4 + 2] * p.scale, acc[rb][d][4 * g4 + 3] * p.scale);
                        *reinterpret_cast<u32x2_t*>(row + (32 * d + 8 * g4 + 4 * hi) * 2) = u;
                    }
            }
        }
    }
}

template <class T, int D>
int launch_dqs(const BwdArgs& a, hipStream_t stream) {
    DqsParams p;
    p.k = a.k; p.ds = reinterpret_cast<const char*>(a.ds); p.dq = a.dq;
    p.B = a.B; p.Hq = a.Hq; p.Hkv = a.Hkv; p.Sq = a.Sq; p.Sk = a.Sk;
    p.scale = a.scale;
    p.coff = a.causal ? a.coff : 0;
    const DsLayout dl = DsLayout::of(a.Hq, a.Hkv, a.Sq, a.Sk);
    p.nq32 = dl.nq32; p.nkb32p = dl.nkb32p;
    const int nqb = (a.Sq + kDqsQBlock - 1) / kDqsQBlock;
    // (half-empty grids -- the cache-sized problems the default dispatch sends here are small ones: when every Q block can have a workgroup slot
    // of its own, the blocks are not paired)
    p.pair = (a.causal && (long long)nqb * a.B * a.Hq > (long long)DQS_OCC * device_cu_count(-1)) ? 1 : 0;
    p.nblk = (a.causal && p.pair) ? (nqb + 1) / 2 : nqb;
    static const int rev = [] {
        const char* e = std::getenv("AULE_HIP_DQS_REV");
        return e != nullptr ? std::atoi(e) : 1;
    }();
    p.rev = rev;
    const dim3 grid((unsigned)(p.nblk * a.B * a.Hq)), block(256);
    if (a.causal)
        hipLaunchKernelGGL((fa_bwd_dqs_kernel<T, D, true>), grid, block, DqsCfg<D>::LDS, stream, p);
    else
        hipLaunchKernelGGL((fa_bwd_dqs_kernel<T, D, false>), grid, block, DqsCfg<D>::LDS, stream, p);
    return (int)hipGetLastError();
}

template <class T, int D>
int launch_delta16(const BwdArgs& a, float* lse2, float* ndelta, hipStream_t stream) {
    Delta16Params p;
    p.o = a.o; p.dout = a.dout; p.lse = a.lse; p.delta = a.delta; p.lse2 = lse2; p.ndelta = ndelta;
    p.rows = (long long)a.B * a.Hq * a.Sq;
    constexpr int RPL = (256 / (D / 8)) * 4;   // rows per workgroup
    hipLaunchKernelGGL((fa_bwd_delta16_kernel<T, D>), dim3((unsigned)((p.rows + RPL - 1) / RPL)), dim3(256), 0, stream, p);
    return (int)hipGetLastError();
}

}  // namespace

// Shapes the 5-matmul backward can take (on top of bwd_dkv4_applicable(), which the dispatcher asks too): the dS column block of
// a (batch, KV head) group and the K rows of the ring's look-ahead inside 32-bit descriptor offsets.
bool bwd_dqs_applicable(const BwdArgs& a) {
    if (a.dtype != kBF16 && a.dtype != kF16) return false;
    if ((a.D != 128 && a.D != 64) || a.window > 0) return false;
    if (a.causal && a.coff < 0) return false;
    if (a.Hkv <= 0 || a.Hq % a.Hkv != 0) return false;
    const DsLayout dl = DsLayout::of(a.Hq, a.Hkv, a.Sq, a.Sk);
    if (dl.group_bytes + (long long)(kDqsAhead + 1) * dl.xs * 2048 >= (1LL << 32)) return false;
    if ((long long)a.Sq * a.D * 2 >= (1LL << 31) || ((long long)a.Sk + (kDqsAhead + 1) * kDqsKB) * a.D * 2 >= (1LL << 31)) return false;
    return true;
}

int launch_bwd_delta16(const BwdArgs& a, float* lse2, float* ndelta, hipStream_t stream) {
    if (a.D == 128) {
        if (a.dtype == kBF16) return launch_delta16<Bf16Traits, 128>(a, lse2, ndelta, stream);
        if (a.dtype == kF16) return launch_delta16<F16Traits, 128>(a, lse2, ndelta, stream);
    } else if (a.D == 64) {
        if (a.dtype == kBF16) return launch_delta16<Bf16Traits, 64>(a, lse2, ndelta, stream);
        if (a.dtype == kF16) return launch_delta16<F16Traits, 64>(a, lse2, ndelta, stream);
    }
    return -1;
}

int launch_bwd_dqs(const BwdArgs& a, hipStream_t stream) {
    if (a.D == 128) {
        if (a.dtype == kBF16) return launch_dqs<Bf16Traits, 128>(a, stream);
        if (a.dtype == kF16) return launch_dqs<F16Traits, 128>(a, stream);
    } else if (a.D == 64) {
        if (a.dtype == kBF16) return launch_dqs<Bf16Traits, 64>(a, stream);
        if (a.dtype == kF16) return launch_dqs<F16Traits, 64>(a, stream);
    }
    return -1;
}

int configure_bwd_dqs() {
    int rc = 0;
    auto set = [&](const void* f, int lds) { rc |= (int)hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, lds); };
    set(reinterpret_cast<const void*>(&fa_bwd_dqs_kernel<Bf16Traits, 128, true>), DqsCfg<128>::LDS);
    set(reinterpret_cast<const void*>(&fa_bwd_dqs_kernel<Bf16Traits, 128, false>), DqsCfg<128>::LDS);
    set(reinterpret_cast<const void*>(&fa_bwd_dqs_kernel<F16Traits, 128, true>), DqsCfg<128>::LDS);
    set(reinterpret_cast<const void*>(&fa_bwd_dqs_kernel<F16Traits, 128, false>), DqsCfg<128>::LDS);
    set(reinterpret_cast<const void*>(&fa_bwd_dqs_kernel<Bf16Traits, 64, true>), DqsCfg<64>::LDS);
    set(reinterpret_cast<const void*>(&fa_bwd_dqs_kernel<Bf16Traits, 64, false>), DqsCfg<64>::LDS);
    set(reinterpret_cast<const void*>(&fa_bwd_dqs_kernel<F16Traits, 64, true>), DqsCfg<64>::LDS);
    set(reinterpret_cast<const void*>(&fa_bwd_dqs_kernel<F16Traits, 64, false>), DqsCfg<64>::LDS);
    return rc;
}

}  // namespace aule_hip
