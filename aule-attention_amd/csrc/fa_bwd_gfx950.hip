// fa_bwd_gfx950.hip -- FlashAttention-2 backward for MI355X (gfx950 / CDNA4).
//
// Replaces, behind aule_attention_backward_ex, the reference's Triton backward
// (python/aule/triton_flash_amd.py:447-500 + kernels :247-380; generic twin
// python/aule/triton_flash.py:478-526, :242-379) and the Vulkan BackwardPipeline
// (src/attention_backward_pipeline.zig:472-537, shaders/attention_backward_f32.comp).
// Math: delta_i = sum_d O_id dO_id ; p_ij = exp(s_ij - LSE_i)
// (0 where masked) ; dV += P^T dO ; dP = dO V^T ; dS = P o (dP - delta) * scale ;
// dQ = dS K ; dK = dS^T Q ; GQA: dK/dV reduce over the query heads of the group.
//
// Structure -- three launches, no atomics, deterministic:
//   1. fa_bwd_delta   : delta = rowsum(O o dO)  (HBM-bound stream)
//   2. fa_bwd_dq      : one workgroup per 256-row Q block (8 waves x 32 rows) loops
//                       over 64-row KV tiles; lane owns one query row, so LSE/delta
//                       are lane-local scalars (same swapped layout as the forward).
//   3. fa_bwd_dkdv    : one workgroup per 128-row KV block (4 waves x 32 rows, one
//                       wave per SIMD, K/V fragments resident in registers) loops
//                       over the group's query heads and 32-row Q tiles; lane owns
//                       one key row, dK^T / dV^T accumulate in registers; the GQA
//                       group reduction is the loop, not an atomic.
// The softmax is recomputed twice (once per kernel): 7 tile matmuls instead of the
// 5 of the atomic formulation, in exchange for no fp32 atomics on dQ.
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <type_traits>

#include "fa_device.h"
#include "fa_kernels.h"

namespace aule_hip {
static bool dkv4_timeline_wanted() { const char* e = std::getenv("AULE_TL"); return e != nullptr && e[0] == 'd' && e[1] == 'k'; }   // AULE_TL=dkv4 (debug library)
bool bwd_dkv4_applicable(const BwdArgs& a);          // fa_bwd_dkv4_gfx950.hip: the one-wave-per-SIMD dK/dV kernel
bool bwd_dkv4_forced();
bool bwd_dkv4_k2(const BwdArgs& a);                    // D = 64: the two-key-blocks-per-wave instance (round 6)
long long bwd_dkv4_items(const BwdArgs& a);
bool bwd_dq4_applicable(const BwdArgs& a);           // fa_bwd_dq4_gfx950.hip: the one-wave-per-SIMD dQ kernel
int bwd_dq4_mode();
int launch_bwd_dq4(const BwdArgs& a, float* lse2_out, float* ndelta_out, hipStream_t stream);
int configure_bwd_dq4();
int launch_bwd_dkv4(const BwdArgs& a, hipStream_t stream);
int configure_bwd_dkv4();
bool bwd_dqs_applicable(const BwdArgs& a);           // fa_bwd_dqs_gfx950.hip: the 5-matmul backward's delta pass and dQ = dS K kernel
int launch_bwd_delta16(const BwdArgs& a, float* lse2, float* ndelta, hipStream_t stream);
int launch_bwd_dqs(const BwdArgs& a, hipStream_t stream);
int configure_bwd_dqs();
uint64_t bwd_f32_partial_bytes(int B, int Hq, int Hkv, int Sq, int Sk, int D, int causal, int device);   // fa_bwd_f32.hip: planes of its small-grid pieces
namespace {

// ------------------------------------------------------------------ delta ----
struct DeltaParams {
    const void* o;
    const void* dout;
    float* delta;
    long long rows;  // B*Hq*Sq
};

// (16-bit I/O: delta is computed inside the dQ kernel, see fa_bwd_dq_kernel; fp32 I/O keeps a separate kernel)
// CPR 16-byte chunks per row; one lane per chunk, CPR-lane groups reduce by shuffle.
template <int D>
__global__ void __launch_bounds__(256) fa_bwd_delta_f32_kernel(const DeltaParams p) {
    constexpr int CPR = D * 4 / 16;  // 8, 16 or 32 lanes per row
    constexpr int RPB = 256 / CPR;
    const int tid = threadIdx.x;
    const int sub = tid % CPR;
    const long long row = (long long)blockIdx.x * RPB + tid / CPR;
    float acc = 0.f;
    if (row < p.rows) {
        const f32x4_t a = reinterpret_cast<const f32x4_t*>(p.o)[row * CPR + sub];
        const f32x4_t b = reinterpret_cast<const f32x4_t*>(p.dout)[row * CPR + sub];
        acc = a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3];
    }
#pragma unroll
    for (int off = CPR / 2; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if (row < p.rows && sub == 0) p.delta[row] = acc;
}

// ------------------------------------------------------------- shared bits ----
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_srd_b(const void* base, unsigned bytes) {
    // raw buffer (stride 0): loads at offsets >= bytes return 0 -> ragged tiles need no clamping
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}

struct BwdParams {
    const void* q;
    const void* k;
    const void* v;
    const void* dout;
    const float* lse;
    const float* delta;   // dkdv kernel: read (written by the dQ kernel)
    const void* o;        // dQ kernel: forward output, for delta = rowsum(O * dO)
    float* delta_out;     // dQ kernel: where it publishes delta for the dK/dV kernel
    float* lse2_out;      // dQ kernel: where it publishes L' = LSE log2(e) (the one-wave-per-SIMD dK/dV kernel reads it scaled)
    float* ndelta_out;    // dQ kernel: ... and - delta (that kernel starts its dP product from it)
    void* dq;
    void* dk;
    void* dv;
    int B, Hq, Hkv, Sq, Sk;
    float c;      // scale * log2(e)   (sign kept; no max is taken in the backward)
    float scale;  // applied to dQ / dK in the epilogue
    int nblk;     // Q blocks (dq kernel) or KV blocks / block pairs (dkdv kernel)
    int gsplit;   // dkdv: the query heads of a GQA group are split over this many workgroups
    float* part;  // dkdv, gsplit > 1: fp32 partials [2 (dK,dV)][gsplit][B,Hkv,Sk,D]
    int window;   // sliding window: key j visible to query i only if i - j < window (0: off)
    int coff;     // causal position offset (query i sits at position i + coff; 0 = top-left rule)
    unsigned long long* dbg;  // timeline build of the dK/dV kernel only: [8 waves][kBwdTLMax] s_memtime stamps of workgroup 0
};

constexpr int kBwdTLMax = 384;   // 6 stamps per tile

#ifndef AULE_DKV_PINNED
#define AULE_DKV_PINNED 1        // 0: the plain-IR P/dS arithmetic (A/B builds)
#endif
constexpr bool kDkvPinned = AULE_DKV_PINNED != 0;
#ifndef AULE_DQ_PINNED
#define AULE_DQ_PINNED 1         // 0: the plain-IR P/dS arithmetic of the dQ kernel (A/B builds)
#endif
constexpr bool kDqPinned = AULE_DQ_PINNED != 0;
#ifndef AULE_DQ_SDP_AHEAD
#define AULE_DQ_SDP_AHEAD 1       // k-steps of operand look-ahead in the dQ kernel's S/dP product.  Without the pinned
                                  // order below 0/1/2 measure the same (hipcc reorders the unrolled block); pinned, 1 is
                                  // -2.5..3 % on the whole causal-D128 backward and 2 is +10 % (92 B/lane of scratch)
#endif
#ifndef AULE_DQ_SDP_PIN
#define AULE_DQ_SDP_PIN 1         // pin the read / MFMA interleave of that block with sched_group_barrier (needs AHEAD > 0)
#endif
#ifndef AULE_DQ_MM_AHEAD
#define AULE_DQ_MM_AHEAD 2         // operand look-ahead of the dQ product (transpose reads)
#endif
#ifndef AULE_DQ_MM_PIN
#define AULE_DQ_MM_PIN 0           // 1: pin its read / MFMA interleave (A/B builds)
#endif
#ifndef AULE_DKV_SCAL_EARLY
#define AULE_DKV_SCAL_EARLY 0      // 1: request the LSE' / delta quads one arithmetic block ahead.  Measured: the phase
                                   // 1140 -> 1080 cycles and the tile 4670 -> 4535, but the whole backward flat within noise in
                                   // three passes, for 12 B/lane of scratch on causal D128 -- off
#endif
constexpr bool kDkvScalEarly = AULE_DKV_SCAL_EARLY != 0;
#ifndef AULE_DKV_MPRIO
#define AULE_DKV_MPRIO 0         // s_setprio level around the dK/dV kernel's MFMA loops (A/B builds)
#endif

// P and dS of four scores (consecutive query rows of one key column) as single-issue instructions in a fixed order:
//   x = S*c - LSE', P = exp2(x), dS = P * (dP - delta), packs of (P0,P1) (P2,P3) (dS0,dS1) (dS2,dS3).
// As plain IR on f32x2 hipcc emits v_pk_fma / v_pk_mul / v_pk_add, the forms that run ~3.6x slower beside the SIMD
// partner's MFMA stream (tools/probe_issue.hip); the dK/dV timeline had this phase at 1050-1900 cycles of a ~5000-cycle
// tile, longer than either MFMA phase.  Inputs, temporaries and outputs are separate operands (29 of the 30 allowed;
// tying them makes hipcc copy the MFMA result tuples); every v_exp result is first read >= 2 instructions later
// (inline asm is invisible to the hazard recogniser).
#define AULE_PDS_QUAD(CVT)                                                                                         \
    asm volatile("v_fma_f32 %4, %12, %28, -%20\n\t"                                                                \
                 "v_fma_f32 %5, %13, %28, -%21\n\t"                                                                \
                 "v_fma_f32 %6, %14, %28, -%22\n\t"                                                                \
                 "v_fma_f32 %7, %15, %28, -%23\n\t"                                                                \
                 "v_exp_f32 %4, %4\n\t"                                                                            \
                 "v_exp_f32 %5, %5\n\t"                                                                            \
                 "v_exp_f32 %6, %6\n\t"                                                                            \
                 "v_exp_f32 %7, %7\n\t"                                                                            \
                 "v_sub_f32 %8, %16, %24\n\t"                                                                      \
                 "v_sub_f32 %9, %17, %25\n\t"                                                                      \
                 "v_sub_f32 %10, %18, %26\n\t"                                                                     \
                 "v_sub_f32 %11, %19, %27\n\t"                                                                     \
                 CVT " %0, %4, %5\n\t"                                                                             \
                 CVT " %1, %6, %7\n\t"                                                                             \
                 "v_mul_f32 %8, %8, %4\n\t"                                                                        \
                 "v_mul_f32 %9, %9, %5\n\t"                                                                        \
                 "v_mul_f32 %10, %10, %6\n\t"                                                                      \
                 "v_mul_f32 %11, %11, %7\n\t"                                                                      \
                 CVT " %2, %8, %9\n\t"                                                                             \
                 CVT " %3, %10, %11\n\t"                                                                           \
                 : "=&v"(p01), "=&v"(p23), "=&v"(d01), "=&v"(d23), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3),     \
                   "=&v"(u0), "=&v"(u1), "=&v"(u2), "=&v"(u3)                                                      \
                 : "v"(s0), "v"(s1), "v"(s2), "v"(s3), "v"(g0), "v"(g1), "v"(g2), "v"(g3), "v"(l0), "v"(l1),        \
                   "v"(l2), "v"(l3), "v"(e0), "v"(e1), "v"(e2), "v"(e3), "v"(c))
template <class T>
__device__ __forceinline__ void pds_quad(float s0, float s1, float s2, float s3, float g0, float g1, float g2, float g3,
                                         float l0, float l1, float l2, float l3, float e0, float e1, float e2, float e3,
                                         float c, unsigned& p01, unsigned& p23, unsigned& d01, unsigned& d23) {
#if defined(__HIP_DEVICE_COMPILE__)
    float t0, t1, t2, t3, u0, u1, u2, u3;
    if constexpr (T::kDType == 2) AULE_PDS_QUAD("v_cvt_pk_bf16_f32");
    else AULE_PDS_QUAD("v_cvt_pk_f16_f32");   // round-to-nearest-even, like F16Traits::pack2
#else
    (void)s0; (void)s1; (void)s2; (void)s3; (void)g0; (void)g1; (void)g2; (void)g3; (void)l0; (void)l1; (void)l2; (void)l3;
    (void)e0; (void)e1; (void)e2; (void)e3; (void)c;
    p01 = p23 = d01 = d23 = 0u;
#endif
}

// The dQ kernel's variant: only dS is needed there, and LSE' / delta are per lane (a lane owns a query row).
#define AULE_DS_PAIR(CVT)                                                                          \
    asm volatile("v_fma_f32 %1, %5, %9, %10\n\t"                                                   \
                 "v_fma_f32 %2, %6, %9, %10\n\t"                                                   \
                 "v_sub_f32 %3, %7, %11\n\t"                                                       \
                 "v_exp_f32 %1, %1\n\t"                                                            \
                 "v_exp_f32 %2, %2\n\t"                                                            \
                 "v_sub_f32 %4, %8, %11\n\t"                                                       \
                 "v_mul_f32 %3, %3, %1\n\t"                                                        \
                 "v_mul_f32 %4, %4, %2\n\t"                                                        \
                 CVT " %0, %3, %4\n\t"                                                             \
                 : "=&v"(d01), "=&v"(t0), "=&v"(t1), "=&v"(u0), "=&v"(u1)                          \
                 : "v"(s0), "v"(s1), "v"(g0), "v"(g1), "v"(c), "v"(nl), "v"(dl))
template <class T>
__device__ __forceinline__ unsigned ds_pair(float s0, float s1, float g0, float g1, float c, float nl, float dl) {
    unsigned d01 = 0u;
#if defined(__HIP_DEVICE_COMPILE__)
    float t0, t1, u0, u1;
    if constexpr (T::kDType == 2) AULE_DS_PAIR("v_cvt_pk_bf16_f32");
    else AULE_DS_PAIR("v_cvt_pk_f16_f32");
#else
    (void)s0; (void)s1; (void)g0; (void)g1; (void)c; (void)nl; (void)dl;
#endif
    return d01;
}

// Row-major image swizzle (shared with the forward's K image).
template <int D>
__device__ __forceinline__ int rswz(int row) {
    if constexpr (D == 128) return row & 15;
    else if constexpr (D == 64) return (row >> 1) & 7;
    else return (row >> 2) & 3;
}
// byte offset of 16-byte chunk cc of row `row` in a row-major swizzled [rows][D] 16-bit image
template <int D>
__device__ __forceinline__ int rm_off(int row, int cc) {
    return row * (D * 2) + ((cc ^ rswz<D>(row)) << 4);
}
// byte offset of the same chunk in the [row/4][D/16][4][16] sub-tiled image (transpose-read source)
template <int D>
__device__ __forceinline__ int st_off(int row, int cc) {
    return ((row >> 2) * (D / 16) + (cc >> 1)) * 128 + (row & 3) * 32 + (cc & 1) * 16;
}

// Write the lane's 4 contiguous-d results of 4 accumulator registers as one 8-byte store
template <class T>
__device__ __forceinline__ void store4(char* row_ptr, int d_elem, float a, float b, float c2, float d2) {
    u32x2_t u;
    u[0] = T::pack2(a, b);
    u[1] = T::pack2(c2, d2);
    *reinterpret_cast<u32x2_t*>(row_ptr + d_elem * 2) = u;
}

// --------------------------------------------------------------- dQ kernel ----
// One workgroup per 256-row Q block (causal: per block PAIR (i, n-1-i), uniform work), 8 waves x 32 query
// rows; lane owns one query row (Q, dO fragments in registers; LSE, delta lane-local scalars).
// Same two-group "ping-pong" schedule as the forward (fa_fwd_pp_gfx950.hip): per 64-row KV tile a
// V-phase (P, dS from S^T, dP^T: VALU only) and an M-phase (dQ^T += K_j^T.dS_j^T, then S^T_{j+1} =
// K_{j+1}.Q^T and dP^T_{j+1} = V_{j+1}.dO^T: 48 MFMAs), wave groups 0-3 / 4-7 one phase apart.
// LDS images per KV tile t: K row-major padded (Krm[t&1]), V row-major padded (Vrm[t&1]) -- both read in
// M-phase(t-1) -- and K sub-tiled (Kst[t%3], transpose-read source, read in M-phase(t); three buffers
// because it lives one tile longer than the row-major images written with it).  Staging rule (all in
// V-phases): in V-phase(tau) group d writes the tile it loaded one phase earlier, tile tau+1+d, then
// requests tile tau+2+d.  Write slot of tile T: 2T-2 (group 0) / 2T-3 (group 1); its buffers' previous
// readers finished in slot 2T-4; its first reader is group 0's M-phase(T-1) in slot 2T-1.
constexpr int kDqQBlock = 256;
constexpr int kDqKV = 64;

// AULE_DQ_DMA=1 (D >= 64): the dQ kernel's K / V images arrive by LDS-DMA (buffer_load ... lds) instead of through staging
// registers and six ds_write_b128 per thread and tile -- as in the round-2 forward (the tile-stream kernel, retired in round 4): un-padded row-major
// images with the 16-byte chunks XOR-swizzled, the sub-tiled image as it was (it is lane-linear).
#ifndef AULE_DQ_DMA
#define AULE_DQ_DMA 1
#endif
template <int D>
struct DqCfg {
    static constexpr int RB = D * 2, RBP = RB + 16, CPR = RB / 16;
    static constexpr bool kDMA = AULE_DQ_DMA != 0 && D >= 64;
    static constexpr int RM = kDqKV * (kDMA ? RB : RBP), ST = kDqKV * RB, NCHUNK = kDqKV * CPR;
    static constexpr int CH = (NCHUNK + 511) / 512, KS = D / 16, DB = D / 32;
    static constexpr bool kFull = (NCHUNK % 512) == 0;
    static constexpr int NVRM = kDMA ? 3 : 2;        // V row-major images (DMA: group 0 requests its half two phases earlier)
    static constexpr int LDS = (2 + NVRM) * RM + 3 * ST;  // Krm x2, Vrm x2 (DMA: x3), Kst x3
};

template <class T, int D, bool CAUSAL, bool TL = false>
__global__ void __launch_bounds__(512) fa_bwd_dq_kernel(const BwdParams p) {
    // TL: s_memtime stamps of workgroup 0, eight per tile (tools/timeline_bwd.py dq):
    //   0 V-phase start  1 next tile written to LDS  2 loads issued  3 dS arithmetic done  4 barrier passed (M-phase start)
    //   5 dQ MFMAs (16) retired  6 S/dP MFMAs of the next tile (32) retired  7 barrier passed
    int tl_n = 0;
    auto stamp = [&]() __attribute__((always_inline)) {
        if constexpr (TL) {
            if (blockIdx.x == 0 && tl_n < kBwdTLMax) {
                const unsigned long long t = __builtin_amdgcn_s_memtime();
                if ((threadIdx.x & 63) == 0) p.dbg[(threadIdx.x >> 6) * kBwdTLMax + tl_n] = t;
                ++tl_n;
            }
        }
    };
    using Cfg = DqCfg<D>;
    using v8 = typename T::v8;
    constexpr int RB = Cfg::RB, RBP = Cfg::RBP, RM = Cfg::RM, ST = Cfg::ST, CH = Cfg::CH, KS = Cfg::KS, DB = Cfg::DB;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr bool DMA = Cfg::kDMA;
    constexpr int NVRM = Cfg::NVRM, CPR = Cfg::CPR;
    char* const Krm = smem;
    char* const Vrm = smem + 2 * RM;
    char* const Kst = smem + (2 + NVRM) * RM;

    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2;
    const WorkItem w = decode_work(blockIdx.x, p.B, p.Hq, p.Hkv, p.nblk, false);
    const int Sq = p.Sq, Sk = p.Sk;
    const float c = p.c;
    const int nqb = (Sq + kDqQBlock - 1) / kDqQBlock;
    const size_t qbase = (size_t)(w.b * p.Hq + w.h) * Sq;
    const size_t kvhead = (size_t)(w.b * p.Hkv + w.hk) * Sk * RB;
    const __amdgpu_buffer_rsrc_t krs = make_srd_b(reinterpret_cast<const char*>(p.k) + kvhead, (unsigned)Sk * RB);
    const __amdgpu_buffer_rsrc_t vrs = make_srd_b(reinterpret_cast<const char*>(p.v) + kvhead, (unsigned)Sk * RB);
    const __amdgpu_buffer_rsrc_t qrs = make_srd_b(reinterpret_cast<const char*>(p.q) + qbase * RB, (unsigned)Sq * RB);
    const __amdgpu_buffer_rsrc_t grs = make_srd_b(reinterpret_cast<const char*>(p.dout) + qbase * RB, (unsigned)Sq * RB);

    // staging map: 8 consecutive lanes fetch one [4 kv][16 d] sub-tile (sub-tiled image filled linearly by
    // thread id); the same K registers also go to the padded row-major image; V uses the same map.
    int st_g[CH], st_rm[CH];
#pragma unroll
    for (int i = 0; i < CH; ++i) {
        const int bidx = (tid >> 3) + 64 * i;
        const int row = (bidx / (D / 16)) * 4 + ((tid >> 1) & 3);
        const int cc = (bidx % (D / 16)) * 2 + (tid & 1);
        st_g[i] = row * RB + cc * 16;
        st_rm[i] = row * RBP + cc * 16;
    }
    constexpr int SWSH = CPR == 16 ? 0 : (CPR == 8 ? 1 : 2);
    // DMA: row l31 of an un-padded image, chunk (2 ks + hi) ^ swz(row) = one XOR with 32 ks on this base (see sdp)
    const int a_base = DMA ? l31 * RB + ((((l31 >> SWSH) & (CPR - 1)) ^ hi) * 16) : l31 * RBP + hi * 16;
    const int tr_off = hi * (D / 16) * 128 + ((lane >> 4) & 1) * 128 + (lane & 15) * 8;

    u32x4_t kst[CH], vst[CH];
    int t_lo = 0;  // first KV tile any row of the current Q block can see (sliding window; else 0)
    // ---- DMA staging.  A wave instruction moves 64 x 16 bytes to a wave-uniform LDS address + lane * 16; wave w of a group
    //      owns pieces 4 i + (w & 3) of a 16-piece (D = 64: 8-piece) image, 4096 source bytes apart in both maps, so one
    //      per-lane offset per map + the scalar offset address them.  Row-major images: position (row r, chunk c') holds
    //      global chunk c' ^ swz(r).  Per tile step (start of V-phase(j)):
    //        group 0: Kst_{j+1} -> Kst[(j+1) % 3], lower half of Vrm_{j+2} -> Vrm[(j+2) % 3]
    //        group 1: Krm_{j+2} -> Krm[j & 1],     upper half of Vrm_{j+2}
    //      (every target's last reader finished at least one phase earlier; Vrm needs its third buffer for group 0's half);
    //      a group waits for its requests at the end of the M-phase that follows, one barrier before their first reader.
    constexpr int KP = DMA ? ST / 4096 : 1;   // pieces per wave and full image
    const int rm_off0 = [&] {
        const int q = (wave & 3) * 64 + lane, r = q / CPR, cs = q % CPR;
        return r * RB + (cs ^ ((r >> SWSH) & (CPR - 1))) * 16;
    }();
    const int st_off0 = [&] {
        const int t = (wave & 3) * 64 + lane, bidx = t >> 3;
        return ((bidx / (D / 16)) * 4 + ((t >> 1) & 3)) * RB + ((bidx % (D / 16)) * 2 + (t & 1)) * 16;
    }();
    auto dma_img = [&](const __amdgpu_buffer_rsrc_t& rs, char* img, int off0, int t, int i0, int i1) __attribute__((always_inline)) {
#if defined(__HIP_DEVICE_COMPILE__)
        using lds_ptr = __attribute__((address_space(3))) void*;
#pragma unroll
        for (int i = 0; i < KP; ++i)
            if (i >= i0 && i < i1)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)(img + (4 * i + (wave & 3)) * 1024), 16, off0,
                                                         (t_lo + t) * kDqKV * RB + i * 4096, 0, 0);
#endif
    };
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
    auto issue_loads = [&](int t) {
#pragma unroll
        for (int i = 0; i < CH; ++i)
            if (Cfg::kFull || tid + 512 * i < Cfg::NCHUNK) {
                kst[i] = __builtin_amdgcn_raw_buffer_load_b128(krs, st_g[i], (t_lo + t) * kDqKV * RB, 0);
                vst[i] = __builtin_amdgcn_raw_buffer_load_b128(vrs, st_g[i], (t_lo + t) * kDqKV * RB, 0);
            }
    };
    auto write_tile = [&](int t) {
        char* krm = Krm + (t & 1) * RM;
        char* vrm = Vrm + (t & 1) * RM;
        char* kst_img = Kst + (t % 3) * ST;
#pragma unroll
        for (int i = 0; i < CH; ++i)
            if (Cfg::kFull || tid + 512 * i < Cfg::NCHUNK) {
                *reinterpret_cast<u32x4_t*>(krm + st_rm[i]) = kst[i];
                *reinterpret_cast<u32x4_t*>(vrm + st_rm[i]) = vst[i];
                *reinterpret_cast<u32x4_t*>(kst_img + tid * 16 + i * 8192) = kst[i];
            }
    };

    const int nparts = (CAUSAL && (nqb - 1 - w.blk) != w.blk) ? 2 : 1;
    for (int part = 0; part < nparts; ++part) {
        const int qb = CAUSAL ? (part == 0 ? nqb - 1 - w.blk : w.blk) : w.blk;
        const int q0w = qb * kDqQBlock + wave * 32;
        const int qrow = q0w + l31;
        const int qr = qrow < Sq ? qrow : Sq - 1;
        const int coff = p.coff;
        const int kv_hi = CAUSAL ? min(Sk, qb * kDqQBlock + kDqQBlock + coff) : Sk;
        t_lo = p.window > 0 ? min(max(0, qb * kDqQBlock + coff - p.window + 1) / kDqKV, (kv_hi + kDqKV - 1) / kDqKV - 1) : 0;

        const int nt_p = (kv_hi + kDqKV - 1) / kDqKV - t_lo;   // (= nt below)
        if constexpr (DMA) {
            // tiles 0 and 1 of the row-major images and tile 0 of the sub-tiled one, with Q / dO / O: one HBM round trip
            if (grp == 0) {
                dma_img(krs, Kst, st_off0, 0, 0, KP);
                dma_img(vrs, Vrm, rm_off0, 0, 0, KP / 2);
                if (nt_p > 1) dma_img(vrs, Vrm + RM, rm_off0, 1, 0, KP / 2);
            } else {
                dma_img(krs, Krm, rm_off0, 0, 0, KP);
                dma_img(vrs, Vrm, rm_off0, 0, KP / 2, KP);
                if (nt_p > 1) {
                    dma_img(krs, Krm + RM, rm_off0, 1, 0, KP);
                    dma_img(vrs, Vrm + RM, rm_off0, 1, KP / 2, KP);
                }
            }
        } else {
            issue_loads(0);
        }
        v8 qf[KS], dof[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            qf[ks] = as_v8<T>(__builtin_amdgcn_raw_buffer_load_b128(qrs, qrow * RB + (2 * ks + hi) * 16, 0, 0));
            dof[ks] = as_v8<T>(__builtin_amdgcn_raw_buffer_load_b128(grs, qrow * RB + (2 * ks + hi) * 16, 0, 0));
        }
        const float nlse2 = -p.lse[qbase + qr] * kLog2e;
        // delta_i = sum_d O[i,d] dO[i,d] (triton_flash.py:353-379), fused here: the lane already holds its half of
        // the dO row; the O row is read once, multiplied in fp32, and the two lane halves are added.  Published for
        // the dK/dV kernel, which runs after this one on the same stream.
        float delta;
        {
            const __amdgpu_buffer_rsrc_t ors = make_srd_b(reinterpret_cast<const char*>(p.o) + qbase * RB, (unsigned)Sq * RB);
            float part = 0.f;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const u32x4_t ov = __builtin_amdgcn_raw_buffer_load_b128(ors, qrow * RB + (2 * ks + hi) * 16, 0, 0);
                const u32x4_t gv = __builtin_bit_cast(u32x4_t, dof[ks]);
#pragma unroll
                for (int i = 0; i < 4; ++i) part += T::lo(ov[i]) * T::lo(gv[i]) + T::hi(ov[i]) * T::hi(gv[i]);
            }
            delta = part + xhalf(part);
            if (hi == 0 && qrow < Sq) {
                p.delta_out[qbase + qrow] = delta;
                p.lse2_out[qbase + qrow] = -nlse2;
                p.ndelta_out[qbase + qrow] = -delta;
            }
        }
        const int kv_lim = CAUSAL ? min(Sk - 1, qrow + coff) : Sk - 1;  // last key visible to this lane's query row

        const int kv_low = p.window > 0 ? qrow + coff - p.window + 1 : -0x40000000;  // first key visible to this lane's row
        const int nt = (kv_hi + kDqKV - 1) / kDqKV - t_lo;
        const int wave_kv_hi = CAUSAL ? min(Sk, q0w + 32 + coff) : Sk;
        const int na = max(1, (wave_kv_hi + kDqKV - 1) / kDqKV - t_lo);

        f32x16_t acc[DB];
#pragma unroll
        for (int d = 0; d < DB; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[d][r] = 0.f;
        f32x16_t s[2], dp[2];
        v8 dsb[2][2];

        auto sdp = [&](int t) {  // S^T = K_t.Q^T ; dP^T = V_t.dO^T
            int kro = (t & 1) * RM + a_base;
            asm volatile("" : "+v"(kro));
            const char* krm = Krm + (DMA ? (t & 1) * RM : kro);
            const char* vrm = Vrm + (DMA ? (t % 3) * RM : kro);
            f32x16_t z;
#pragma unroll
            for (int r = 0; r < 16; ++r) z[r] = 0.f;
            constexpr int kAhead = AULE_DQ_SDP_AHEAD;
            u32x4_t ka[KS][2], va[KS][2];
            auto rd = [&](int ks) {
                if constexpr (DMA) {
                    const int a = a_base ^ (ks * 32);   // swizzled chunk of this lane's row (rows +32: same swizzle)
#pragma unroll
                    for (int sb = 0; sb < 2; ++sb) {
                        ka[ks][sb] = *reinterpret_cast<const u32x4_t*>(krm + a + sb * 32 * RB);
                        va[ks][sb] = *reinterpret_cast<const u32x4_t*>(vrm + a + sb * 32 * RB);
                    }
                } else {
#pragma unroll
                    for (int sb = 0; sb < 2; ++sb) {
                        ka[ks][sb] = *reinterpret_cast<const u32x4_t*>(krm + sb * 32 * RBP + ks * 32);
                        va[ks][sb] = *reinterpret_cast<const u32x4_t*>(vrm + sb * 32 * RBP + ks * 32);
                    }
                }
            };
#pragma unroll
            for (int ks = 0; ks < kAhead && ks < KS; ++ks) rd(ks);
            if constexpr (AULE_DQ_SDP_PIN != 0 && kAhead > 0)
                __builtin_amdgcn_sched_group_barrier(0x100, 4 * (kAhead < KS ? kAhead : KS), 0);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                if (kAhead == 0) rd(ks);
                else if (ks + kAhead < KS) rd(ks + kAhead);
#pragma unroll
                for (int sb = 0; sb < 2; ++sb) {
                    s[sb] = T::mfma(as_v8<T>(ka[ks][sb]), qf[ks], ks == 0 ? z : s[sb]);
                    dp[sb] = T::mfma(as_v8<T>(va[ks][sb]), dof[ks], ks == 0 ? z : dp[sb]);
                }
                if constexpr (AULE_DQ_SDP_PIN != 0 && kAhead > 0) {   // order pinned: next step's reads, then this step's MFMAs
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    if (ks + kAhead < KS) __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
                }
            }
        };
        auto dq_mm = [&](int t) {  // dQ^T += K_t^T . dS_t^T  (A = K^T by transpose read, B = dS in registers)
            int kofs = (2 + NVRM) * RM + (t % 3) * ST + tr_off;
            asm volatile("" : "+v"(kofs));  // one base register + 16-bit immediates (else 32 hoisted addresses spill)
            const char* ktr = smem + kofs;
            constexpr int NST = 4 * DB, kAhead = AULE_DQ_MM_AHEAD;
            s16x4_t a0[NST], a1[NST];
            auto rd = [&](int st) {
                const int sk = st / DB, d = st % DB;  // sk = 2*sb + kk
                const int off = ((4 * sk) * (D / 16) + 2 * d) * 128;
                a0[st] = lds_tr16(ktr + off);
                a1[st] = lds_tr16(ktr + off + 2 * (D / 16) * 128);
            };
#pragma unroll
            for (int st = 0; st < kAhead && st < NST; ++st) rd(st);
            if constexpr (AULE_DQ_MM_PIN != 0) __builtin_amdgcn_sched_group_barrier(0x100, 2 * (kAhead < NST ? kAhead : NST), 0);
#pragma unroll
            for (int st = 0; st < NST; ++st) {
                if (st + kAhead < NST) rd(st + kAhead);
                const int sk = st / DB, d = st % DB;
                acc[d] = T::mfma(as_v8<T>(a0[st], a1[st]), dsb[sk >> 1][sk & 1], acc[d]);
                if constexpr (AULE_DQ_MM_PIN != 0) {   // order pinned: one MFMA, then the two transpose reads of step st + kAhead
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    if (st + kAhead < NST) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                }
            }
        };
        auto softmax = [&](int kv0) {  // P^T = exp2(S^T c - LSE log2e) (0 where masked) ; dS^T = P^T o (dP^T - delta)
            const bool need_mask = (CAUSAL && (kv0 + kDqKV - 1 > q0w + coff)) || (kv0 + kDqKV > Sk) ||
                                   (p.window > 0 && q0w + coff + 31 - kv0 >= p.window);
            const f32x2_t c2 = {c, c}, nl2 = {nlse2, nlse2}, dl2 = {delta, delta};
            int rel = kv0 - kv_lim;  // key index relative to the last visible key of this lane's row
            int rlo = kv0 - kv_low;  // ... and to the first visible one (sliding window)
            asm volatile("" : "+v"(rel), "+v"(rlo));  // (opaque: otherwise hipcc hoists 32 per-element constants out of the loop and spills them)
            u32x4_t du[2][2];
            if (kDqPinned && !need_mask) {   // (wave-uniform) steady state: pinned single-issue form, two scores a statement
#pragma unroll
                for (int sb = 0; sb < 2; ++sb)
#pragma unroll
                    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const int r = 8 * kk + 2 * j;
                            du[sb][kk][j] = ds_pair<T>(s[sb][r], s[sb][r + 1], dp[sb][r], dp[sb][r + 1], c, nlse2, delta);
                        }
            } else
#pragma unroll
            for (int sb = 0; sb < 2; ++sb)
#pragma unroll
                for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int r = 8 * kk + 2 * j;
                        f32x2_t t2 = {s[sb][r], s[sb][r + 1]};
                        t2 = __builtin_elementwise_fma(t2, c2, nl2);
                        t2[0] = fast_exp2(t2[0]);
                        t2[1] = fast_exp2(t2[1]);
                        if (need_mask) {  // visible iff kv <= kv_lim (one compare per element, no branches)
                            const int kvr = rel + sb * 32 + crow(r, hi);
                            const int kvl = rlo + sb * 32 + crow(r, hi);
                            t2[0] = (kvr <= 0 && kvl >= 0) ? t2[0] : 0.f;
                            t2[1] = (kvr < 0 && kvl >= -1) ? t2[1] : 0.f;
                        }
                        const f32x2_t dpv = {dp[sb][r], dp[sb][r + 1]};
                        const f32x2_t dsv = t2 * (dpv - dl2);
                        du[sb][kk][j] = T::pack2(dsv[0], dsv[1]);
                    }
            // pin the phase's results here (see fa_fwd_pp_gfx950.hip: hipcc sinks register-only code past barriers)
            asm volatile("" : "+v"(du[0][0]), "+v"(du[0][1]), "+v"(du[1][0]), "+v"(du[1][1]));
#pragma unroll
            for (int sb = 0; sb < 2; ++sb)
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) dsb[sb][kk] = as_v8<T>(du[sb][kk]);
        };

        // ---- prologue: tile 0 -> all images (all waves); group 0 holds tile 1 in registers, group 1 writes
        //      its share of tile 1 and holds tile 2.
        if constexpr (DMA) {
            phase_barrier(true);            // the part's first tiles (requested above) have landed
            if (grp == 1) phase_barrier(false);
        } else {
            write_tile(0);
            if (nt > 1) issue_loads(1);
            if (grp == 1) {
                if (nt > 1) write_tile(1);
                if (nt > 2) issue_loads(2);
            }
            __syncthreads();
            if (grp == 1) __syncthreads();  // group 1 starts one phase late
        }
        if (na > 0) sdp(0);             // pre-phase
        phase_barrier(false);

        auto tile_step = [&](int j, auto mode_tag) {
            constexpr int MODE = decltype(mode_tag)::value;
            // ---- V-phase(j): staging, then P/dS of tile j
            stamp();   // 0
            if constexpr (DMA) {
                if (grp == 0) {
                    if (j + 1 < nt) dma_img(krs, Kst + ((j + 1) % 3) * ST, st_off0, j + 1, 0, KP);
                    if (j + 2 < nt) dma_img(vrs, Vrm + ((j + 2) % 3) * RM, rm_off0, j + 2, 0, KP / 2);
                } else {
                    if (j + 2 < nt) {
                        dma_img(krs, Krm + (j & 1) * RM, rm_off0, j + 2, 0, KP);
                        dma_img(vrs, Vrm + ((j + 2) % 3) * RM, rm_off0, j + 2, KP / 2, KP);
                    }
                }
                stamp();   // 1
            } else {
                if (j + 1 + grp < nt) write_tile(j + 1 + grp);
                stamp();   // 1
                if (j + 2 + grp < nt) issue_loads(j + 2 + grp);
            }
            stamp();   // 2
            if constexpr (MODE >= 1) softmax((t_lo + j) * kDqKV);
            stamp();   // 3
            phase_barrier(false);
            stamp();   // 4
            // ---- M-phase(j): dQ^T += K_j^T.dS_j^T ; S^T_{j+1}, dP^T_{j+1}
            __builtin_amdgcn_s_setprio(1);
            if constexpr (MODE >= 1) dq_mm(j);
            if constexpr (TL) asm volatile("s_nop 0" : "+v"(acc[0]), "+v"(acc[DB - 1]));
            stamp();   // 5
            if constexpr (MODE == 2) {
                __builtin_amdgcn_sched_barrier(0);
                sdp(j + 1);
            }
            if constexpr (TL) asm volatile("s_nop 0" : "+v"(s[1]), "+v"(dp[1]));
            stamp();   // 6
            __builtin_amdgcn_s_setprio(0);
            phase_barrier(true);
            stamp();   // 7
        };
        int j = 0;
        for (; j + 1 < na; ++j) tile_step(j, std::integral_constant<int, 2>{});
        if (j < na) { tile_step(j, std::integral_constant<int, 1>{}); ++j; }
        for (; j < nt; ++j) tile_step(j, std::integral_constant<int, 0>{});

        if (qrow < Sq) {
            char* orow = reinterpret_cast<char*>(p.dq) + (qbase + qrow) * RB;
            const float sc = p.scale;
#pragma unroll
            for (int d = 0; d < DB; ++d)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4)
                    store4<T>(orow, 32 * d + 8 * g4 + 4 * hi, acc[d][4 * g4] * sc, acc[d][4 * g4 + 1] * sc,
                              acc[d][4 * g4 + 2] * sc, acc[d][4 * g4 + 3] * sc);
        }
        if (grp == 0) phase_barrier(false);  // pairs with group 1's last phase barrier
    }
}

// ------------------------------------------------------------ dK/dV kernel ----
// One workgroup per PAIR of 256-row KV blocks (i, n-1-i) under a causal mask (uniform work per
// workgroup), else per block; 8 waves x 32 key rows = two waves per SIMD, so one wave's LDS latency and
// VALU work hide under its partner's MFMAs (a 4-wave, one-wave-per-SIMD version with K and V both in
// registers ran at 27 % of this kernel's MFMA rate: everything it did was exposed).  To fit the
// 256-register budget of two waves per SIMD next to the 128 accumulator registers (dK^T, dV^T), only the
// K fragments stay in registers; V lives in a per-wave LDS slab (padded rows) and is re-read per tile as
// the B operand of dP = dO.V^T.  Query tiles of 32 rows are double-buffered in LDS in two images each
// (Q and dO): row-major with rows padded by 16 B (A operand of S = Q.K^T and dP, conflict-free
// ds_read_b128, immediate offsets) and [q/4][d/16][4][16] sub-tiles (transpose-read source for
// dV^T += dO^T.P and dK^T += Q^T.dS).
constexpr int kKvBlock = 256;  // 8 waves x 32 key rows
constexpr int kQT = 32;        // query rows per tile
#ifndef AULE_DKV_AHEAD
#define AULE_DKV_AHEAD 1
#endif
constexpr int kDkvAhead = AULE_DKV_AHEAD;  // operand look-ahead of the dK/dV kernel's MFMA loops (steps)

// AULE_DKV_DMA=1: the Q / dO tiles of the dK/dV kernel arrive by LDS-DMA (as in the forward and the dQ kernel): the
// row-major images un-padded with the XOR chunk swizzle, the sub-tiled images as they were; one piece per wave and image.
#ifndef AULE_DKV_DMA
#define AULE_DKV_DMA 1
#endif
template <int D>
struct DkvCfg {
    static constexpr int RB = D * 2, RBP = RB + 16, CPR = RB / 16;
    static constexpr bool kDMA = AULE_DKV_DMA != 0;
    static constexpr int RM = kQT * (kDMA ? RB : RBP);   // row-major image (register staging: rows padded by 16 B)
    static constexpr int ST = kQT * RB;                  // sub-tiled image
    static constexpr int NCHUNK = kQT * CPR;             // <= 512: at most one chunk per thread and tensor
    static constexpr int KS = D / 16, DB = D / 32;
    static constexpr int VSLAB = 32 * RBP;               // one wave's V rows
    // per stage: Q rm, Q st, dO rm, dO st, LSE*log2e[32], delta[32]
    static constexpr int STAGE = 2 * RM + 2 * ST + 256;
    static constexpr int LDS = 8 * VSLAB + 2 * STAGE;
};

template <class T, int D, bool CAUSAL, bool TL = false>
__global__ void __launch_bounds__(512) fa_bwd_dkdv_kernel(const BwdParams p) {
    using Cfg = DkvCfg<D>;
    // TL: s_memtime stamps of workgroup 0 at the phase boundaries of every tile (tools/timeline_bwd.py):
    //   0 loop top (next tile's loads issued)   1 S / dP MFMAs retired   2 P / dS arithmetic done
    //   3 dV / dK MFMAs retired                 4 next tile written to LDS   5 barrier passed
    int tl_n = 0;
    auto stamp = [&]() __attribute__((always_inline)) {
        if constexpr (TL) {
            if (blockIdx.x == 0 && tl_n < kBwdTLMax) {
                const unsigned long long t = __builtin_amdgcn_s_memtime();
                if ((threadIdx.x & 63) == 0) p.dbg[(threadIdx.x >> 6) * kBwdTLMax + tl_n] = t;
                ++tl_n;
            }
        }
    };
    auto retire = [&](f32x16_t& a, f32x16_t& b) __attribute__((always_inline)) {   // make the MFMA results "used" before a stamp
        if constexpr (TL) asm volatile("s_nop 0" : "+v"(a), "+v"(b));
    };
    using v8 = typename T::v8;
    constexpr int RB = Cfg::RB, RBP = Cfg::RBP, CPR = Cfg::CPR, RM = Cfg::RM, ST = Cfg::ST;
    constexpr int KS = Cfg::KS, DB = Cfg::DB, STAGE = Cfg::STAGE;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    char* const Vslab = smem + wave * Cfg::VSLAB;
    char* const stage0 = smem + 8 * Cfg::VSLAB;
    const int g = p.Hq / p.Hkv;
    const int nkb = (p.Sk + kKvBlock - 1) / kKvBlock;
    // one work item per (batch, kv head, kv block or block pair, head-split index): decode_work with
    // Hq := Hkv * gsplit, so that `h - hk*gsplit` is the split index
    const WorkItem w = decode_work(blockIdx.x, p.B, p.Hkv * p.gsplit, p.Hkv, p.nblk, false);
    const int si = w.h - w.hk * p.gsplit;
    const int gh = g / p.gsplit;  // query heads handled by this workgroup: [si*gh, (si+1)*gh)
    const int Sq = p.Sq, Sk = p.Sk;
    const float c = p.c;
    const size_t kvbase = (size_t)(w.b * p.Hkv + w.hk) * Sk;
    const bool stager = Cfg::NCHUNK == 512 || tid < Cfg::NCHUNK;  // wave-uniform (NCHUNK is a multiple of 64)

    // staging map: 8 consecutive lanes fetch one [4 q][16 d] sub-tile, so the sub-tiled image is filled
    // linearly by thread id; the same registers are also written to the padded row-major image.
    const int bidx = tid >> 3;  // sub-tile index = q4 * (D/16) + d16
    const int st_row = (bidx / (D / 16)) * 4 + ((tid >> 1) & 3);
    const int st_cc = (bidx % (D / 16)) * 2 + (tid & 1);
    const int st_g = st_row * RB + st_cc * 16;
    const int st_rm = st_row * RBP + st_cc * 16;
    constexpr bool DMA = Cfg::kDMA;
    constexpr int SWSH = CPR == 16 ? 0 : (CPR == 8 ? 1 : 2);
    const int a_base = l31 * RBP + hi * 16;  // operand rows l31, chunk 2ks + hi (row-major padded images: the V slab, and Q / dO without DMA)
    const int a_sw = l31 * RB + ((((l31 >> SWSH) & (CPR - 1)) ^ hi) * 16);   // DMA images: chunk (2 ks + hi) ^ swz(row) = a_sw ^ 32 ks
    const int tr_off = hi * (D / 16) * 128 + ((lane >> 4) & 1) * 128 + (lane & 15) * 8;
    // DMA: wave w moves piece w (64 x 16 bytes, lane-linear in LDS) of each of the four images of a tile
    constexpr int NPIECE = ST / 1024;   // pieces per image (D = 128: 8 -- one per wave)
    const int rm_g = [&] {
        const int q = wave * 64 + lane, r = q / CPR, cs = q % CPR;
        return r * RB + (cs ^ ((r >> SWSH) & (CPR - 1))) * 16;
    }();
    (void)NPIECE; (void)rm_g;   // (used in the device pass only)

    const int nparts = (CAUSAL && (nkb - 1 - w.blk) != w.blk) ? 2 : 1;
    for (int part = 0; part < nparts; ++part) {
        const int kb = CAUSAL ? (part == 0 ? w.blk : nkb - 1 - w.blk) : w.blk;
        const int n0w = kb * kKvBlock + wave * 32;  // first key row of this wave
        const int kvrow = n0w + l31;

        v8 kf[KS];  // B operand of S: lane (kv, hi) holds d = 16ks + 8hi .. +7 (rows >= Sk read as 0)
        {
            const __amdgpu_buffer_rsrc_t krs = make_srd_b(reinterpret_cast<const char*>(p.k) + kvbase * RB, (unsigned)Sk * RB);
            const __amdgpu_buffer_rsrc_t vrs = make_srd_b(reinterpret_cast<const char*>(p.v) + kvbase * RB, (unsigned)Sk * RB);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
                kf[ks] = as_v8<T>(__builtin_amdgcn_raw_buffer_load_b128(krs, kvrow * RB + (2 * ks + hi) * 16, 0, 0));
            // this wave's 32 V rows -> its LDS slab (wave-private: program order suffices, no barrier)
#pragma unroll
            for (int i = 0; i < (32 * CPR) / 64; ++i) {
                const int cidx = lane + 64 * i;
                const int row = cidx / CPR, cc = cidx % CPR;
                const u32x4_t x = __builtin_amdgcn_raw_buffer_load_b128(vrs, row * RB + cc * 16, n0w * RB, 0);
                *reinterpret_cast<u32x4_t*>(Vslab + row * RBP + cc * 16) = x;
            }
        }

        // query tiles that can see this KV block (top-left causal: q >= kv)
        // (sliding window: and q - kv < window, i.e. q <= last key of the block + window - 1)
        const int W = p.window;
        const int coff = p.coff;   // query q sits at position q + coff
        int ntq_all = (Sq + kQT - 1) / kQT;
        if (W > 0) ntq_all = min(ntq_all, max(0, kb * kKvBlock + kKvBlock - 1 + W - coff + kQT - 1) / kQT);
        const int first_qt = CAUSAL ? max(0, kb * kKvBlock - coff) / kQT : 0;
        const int ntq = ntq_all > first_qt ? ntq_all - first_qt : 0;
        const int nit = ntq * gh;  // flattened (group head, q tile) loop

        u32x4_t qst, dst;
        float sc_st = 0.f;  // staged LSE*log2e (threads 0..31) or delta (threads 32..63)
        auto issue_loads = [&](int it) {
            const int hh = si * gh + it / ntq, qt = first_qt + it % ntq;
            const size_t qb = (size_t)(w.b * p.Hq + w.hk * g + hh) * Sq;
            const __amdgpu_buffer_rsrc_t qrs = make_srd_b(reinterpret_cast<const char*>(p.q) + qb * RB, (unsigned)Sq * RB);
            const __amdgpu_buffer_rsrc_t grs = make_srd_b(reinterpret_cast<const char*>(p.dout) + qb * RB, (unsigned)Sq * RB);
            const int q0 = qt * kQT;
            if constexpr (DMA) {
#if defined(__HIP_DEVICE_COMPILE__)
                // tile `it` -> stage buffer it & 1 (its last readers finished before the barrier that ended iteration it - 2)
                using lds_ptr = __attribute__((address_space(3))) void*;
                if (wave < NPIECE) {
                    char* base = stage0 + (it & 1) * STAGE + wave * 1024;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(qrs, (lds_ptr)(base), 16, rm_g, q0 * RB, 0, 0);
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(qrs, (lds_ptr)(base + RM), 16, st_g, q0 * RB, 0, 0);
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(grs, (lds_ptr)(base + RM + ST), 16, rm_g, q0 * RB, 0, 0);
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(grs, (lds_ptr)(base + 2 * RM + ST), 16, st_g, q0 * RB, 0, 0);
                }
#endif
            } else if (stager) {
                qst = __builtin_amdgcn_raw_buffer_load_b128(qrs, st_g, q0 * RB, 0);
                dst = __builtin_amdgcn_raw_buffer_load_b128(grs, st_g, q0 * RB, 0);
            }
            if (tid < 64) {
                int r = q0 + (tid & 31);
                r = r < Sq ? r : Sq - 1;
                // raw value: scaling the LSE here made the compiler wait for this load inside the loop top -- a full
                // memory round trip (~760 cycles per tile) on wave 0, which the other waves then wait for at the
                // barrier (tools/timeline_bwd.py); it is scaled when it is written to LDS, a tile later
                sc_st = tid < 32 ? p.lse[qb + r] : p.delta[qb + r];
            }
        };
        auto write_stage = [&](int buf) {
            char* base = stage0 + buf * STAGE;
            if (!DMA && stager) {
                *reinterpret_cast<u32x4_t*>(base + st_rm) = qst;
                *reinterpret_cast<u32x4_t*>(base + RM + tid * 16) = qst;
                *reinterpret_cast<u32x4_t*>(base + RM + ST + st_rm) = dst;
                *reinterpret_cast<u32x4_t*>(base + 2 * RM + ST + tid * 16) = dst;
            }
            if (tid < 64) reinterpret_cast<float*>(base + 2 * RM + 2 * ST)[tid] = tid < 32 ? sc_st * kLog2e : sc_st;
        };

        f32x16_t dk[DB], dv[DB];
#pragma unroll
        for (int d = 0; d < DB; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r) { dk[d][r] = 0.f; dv[d][r] = 0.f; }

        if (nit > 0) {
            issue_loads(0);
            write_stage(0);
        }
        if constexpr (DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();

        for (int it = 0; it < nit; ++it) {
            const int cur = it & 1;
            const int q0 = (first_qt + it % ntq) * kQT;
            if (it + 1 < nit) issue_loads(it + 1);
            stamp();   // 0
            // the tile contributes to this wave's keys iff some query row q >= key row exists
            if ((!CAUSAL || q0 + coff + kQT - 1 >= n0w) && (W <= 0 || q0 + coff < n0w + 31 + W)) {
                const char* base = stage0 + cur * STAGE;
                const char* qrm = base + (DMA ? 0 : a_base);
                const char* qtr = base + RM + tr_off;
                const char* grm = base + RM + ST + (DMA ? 0 : a_base);
                const char* gtr = base + 2 * RM + ST + tr_off;
                const char* vsl = Vslab + a_base;
                const float* scal = reinterpret_cast<const float*>(base + 2 * RM + 2 * ST);

                f32x16_t s, dp, z;
#pragma unroll
                for (int r = 0; r < 16; ++r) z[r] = 0.f;
                {   // operands requested kDkvAhead k-steps early (hand-pipelined like the forward's M-phase loops)
                    u32x4_t qa[KS], da[KS], vb[KS];
                    auto rd = [&](int ks) __attribute__((always_inline)) {
                        const int a = DMA ? (a_sw ^ (ks * 32)) : ks * 32;
                        qa[ks] = *reinterpret_cast<const u32x4_t*>(qrm + a);
                        da[ks] = *reinterpret_cast<const u32x4_t*>(grm + a);
                        vb[ks] = *reinterpret_cast<const u32x4_t*>(vsl + ks * 32);
                    };
#pragma unroll
                    for (int ks = 0; ks < kDkvAhead && ks < KS; ++ks) rd(ks);
                    if constexpr (AULE_DKV_MPRIO != 0) __builtin_amdgcn_s_setprio(AULE_DKV_MPRIO);
                    __builtin_amdgcn_sched_group_barrier(0x100, 3 * (kDkvAhead < KS ? kDkvAhead : KS), 0);
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) {
                        if (ks + kDkvAhead < KS) rd(ks + kDkvAhead);
                        s = T::mfma(as_v8<T>(qa[ks]), kf[ks], ks == 0 ? z : s);                  // S  = Q  . K^T
                        dp = T::mfma(as_v8<T>(da[ks]), as_v8<T>(vb[ks]), ks == 0 ? z : dp);      // dP = dO . V^T
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        if (ks + kDkvAhead < KS) __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    }
                }
                if constexpr (AULE_DKV_MPRIO != 0) __builtin_amdgcn_s_setprio(0);
                // LSE' and delta of the tile's query rows, one block ahead: inside the (kk, g4) loop each pair of reads
                // was issued right before its block and waited for with lgkmcnt(0) -- four exposed LDS round trips per
                // tile (the masked path's branches keep hipcc from hoisting them).  The first pair now overlaps the tail
                // of the S / dP MFMAs and pair i + 1 is requested before block i runs.  (All eight at once is too many
                // registers: the D128 kernels went from 253 VGPRs / no scratch to 256 + 72 B/lane.)
                auto scal_rd = [&](int i, f32x4_t& l, f32x4_t& d) __attribute__((always_inline)) {
                    const int r0 = 8 * (i >> 1) + 4 * (i & 1);
                    l = *reinterpret_cast<const f32x4_t*>(scal + 2 * r0 + 4 * hi);
                    d = *reinterpret_cast<const f32x4_t*>(scal + 32 + 2 * r0 + 4 * hi);
                };
                f32x4_t l4n = {0.f, 0.f, 0.f, 0.f}, d4n = {0.f, 0.f, 0.f, 0.f};
                if constexpr (kDkvScalEarly) scal_rd(0, l4n, d4n);
                retire(s, dp);
                stamp();   // 1
                const bool need_mask = (CAUSAL && (q0 + coff < n0w + 31)) || (q0 + kQT > Sq) || (n0w + 32 > Sk) ||
                                       (W > 0 && q0 + coff + kQT - 1 - n0w >= W);
                const f32x2_t c2 = {c, c};
                v8 pb[2], dsb[2];
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    u32x4_t pu, du;
#pragma unroll
                    for (int g4 = 0; g4 < 2; ++g4) {
                        const int r0 = 8 * kk + 4 * g4;  // registers r0..r0+3 = 4 consecutive query rows
                        f32x4_t l4, d4;
                        if constexpr (kDkvScalEarly) {
                            l4 = l4n; d4 = d4n;
                            if (2 * kk + g4 < 3) scal_rd(2 * kk + g4 + 1, l4n, d4n);
                        } else {
                            l4 = *reinterpret_cast<const f32x4_t*>(scal + 2 * r0 + 4 * hi);
                            d4 = *reinterpret_cast<const f32x4_t*>(scal + 32 + 2 * r0 + 4 * hi);
                        }
                        if (kDkvPinned && !need_mask) {   // (wave-uniform) steady state: the pinned single-issue form
                            unsigned p01, p23, d01, d23;   // (a vector element cannot bind to the asm's reference)
                            pds_quad<T>(s[r0], s[r0 + 1], s[r0 + 2], s[r0 + 3], dp[r0], dp[r0 + 1], dp[r0 + 2], dp[r0 + 3],
                                        l4[0], l4[1], l4[2], l4[3], d4[0], d4[1], d4[2], d4[3], c, p01, p23, d01, d23);
                            pu[2 * g4] = p01; pu[2 * g4 + 1] = p23;
                            du[2 * g4] = d01; du[2 * g4 + 1] = d23;
                            continue;
                        }
#pragma unroll
                        for (int j2 = 0; j2 < 2; ++j2) {
                            const int r = r0 + 2 * j2;
                            f32x2_t t = {s[r], s[r + 1]};
                            const f32x2_t nl = {-l4[2 * j2], -l4[2 * j2 + 1]};
                            t = __builtin_elementwise_fma(t, c2, nl);
                            t[0] = fast_exp2(t[0]);
                            t[1] = fast_exp2(t[1]);
                            if (need_mask) {
                                const int q = q0 + crow(r, hi);
                                const bool okc = kvrow < Sk;
                                const int qp = q + coff;
                                t[0] = (okc && q < Sq && (!CAUSAL || kvrow <= qp) && (W <= 0 || qp - kvrow < W)) ? t[0] : 0.f;
                                t[1] = (okc && q + 1 < Sq && (!CAUSAL || kvrow <= qp + 1) && (W <= 0 || qp + 1 - kvrow < W)) ? t[1] : 0.f;
                            }
                            const f32x2_t dpv = {dp[r] - d4[2 * j2], dp[r + 1] - d4[2 * j2 + 1]};
                            const f32x2_t dsv = t * dpv;
                            pu[2 * g4 + j2] = T::pack2(t[0], t[1]);
                            du[2 * g4 + j2] = T::pack2(dsv[0], dsv[1]);
                        }
                    }
                    pb[kk] = as_v8<T>(pu);
                    dsb[kk] = as_v8<T>(du);
                }
                if constexpr (TL) asm volatile("" : "+v"(pb[0]), "+v"(pb[1]), "+v"(dsb[0]), "+v"(dsb[1]));
                stamp();   // 2
                // dV^T += dO^T . P ; dK^T += Q^T . dS   (A by transpose read, k-slot = query row)
                {
                    constexpr int NST = 2 * DB;   // step = (kk, d): one dV and one dK MFMA
                    s16x4_t x0[NST], x1[NST], y0[NST], y1[NST];
                    auto rd = [&](int st) __attribute__((always_inline)) {
                        const int kk = st / DB, d = st % DB;
                        const int off = ((4 * kk) * (D / 16) + 2 * d) * 128;
                        x0[st] = lds_tr16(gtr + off);
                        x1[st] = lds_tr16(gtr + off + 2 * (D / 16) * 128);
                        y0[st] = lds_tr16(qtr + off);
                        y1[st] = lds_tr16(qtr + off + 2 * (D / 16) * 128);
                    };
#pragma unroll
                    for (int st = 0; st < kDkvAhead && st < NST; ++st) rd(st);
                    if constexpr (AULE_DKV_MPRIO != 0) __builtin_amdgcn_s_setprio(AULE_DKV_MPRIO);
                    __builtin_amdgcn_sched_group_barrier(0x100, 4 * (kDkvAhead < NST ? kDkvAhead : NST), 0);
#pragma unroll
                    for (int st = 0; st < NST; ++st) {
                        if (st + kDkvAhead < NST) rd(st + kDkvAhead);
                        const int kk = st / DB, d = st % DB;
                        dv[d] = T::mfma(as_v8<T>(x0[st], x1[st]), pb[kk], dv[d]);
                        dk[d] = T::mfma(as_v8<T>(y0[st], y1[st]), dsb[kk], dk[d]);
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        if (st + kDkvAhead < NST) __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    }
                }
            }
            else {
                stamp(); stamp();   // (tile skipped by this wave: keep six stamps per tile)
            }
            if constexpr (AULE_DKV_MPRIO != 0) __builtin_amdgcn_s_setprio(0);
            retire(dv[DB - 1], dk[DB - 1]);
            stamp();   // 3
            if (it + 1 < nit) write_stage(cur ^ 1);
            stamp();   // 4
            if constexpr (DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the next tile's images have landed
            __syncthreads();
            stamp();   // 5
        }

        if (kvrow < Sk) {
            const float sc = p.scale;
            if (p.gsplit == 1) {
                char* krow = reinterpret_cast<char*>(p.dk) + (kvbase + kvrow) * RB;
                char* vrow = reinterpret_cast<char*>(p.dv) + (kvbase + kvrow) * RB;
#pragma unroll
                for (int d = 0; d < DB; ++d)
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        const int de = 32 * d + 8 * g4 + 4 * hi;
                        store4<T>(krow, de, dk[d][4 * g4] * sc, dk[d][4 * g4 + 1] * sc, dk[d][4 * g4 + 2] * sc,
                                  dk[d][4 * g4 + 3] * sc);
                        store4<T>(vrow, de, dv[d][4 * g4], dv[d][4 * g4 + 1], dv[d][4 * g4 + 2], dv[d][4 * g4 + 3]);
                    }
            } else {  // fp32 partials, summed in a fixed order by fa_bwd_reduce_kernel (deterministic)
                const size_t tensor = (size_t)p.B * p.Hkv * Sk * D;
                float* kp = p.part + (size_t)si * tensor + (kvbase + kvrow) * D;
                float* vp = p.part + ((size_t)p.gsplit + si) * tensor + (kvbase + kvrow) * D;
#pragma unroll
                for (int d = 0; d < DB; ++d)
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        const int de = 32 * d + 8 * g4 + 4 * hi;
                        const f32x4_t a = {dk[d][4 * g4] * sc, dk[d][4 * g4 + 1] * sc, dk[d][4 * g4 + 2] * sc,
                                           dk[d][4 * g4 + 3] * sc};
                        const f32x4_t b = {dv[d][4 * g4], dv[d][4 * g4 + 1], dv[d][4 * g4 + 2], dv[d][4 * g4 + 3]};
                        *reinterpret_cast<f32x4_t*>(kp + de) = a;
                        *reinterpret_cast<f32x4_t*>(vp + de) = b;
                    }
            }
        }
    }
}

// dK/dV = sum over the head-split partials (fixed order), cast to the storage dtype.  HBM-bound.
struct ReduceParams {
    const float* part;
    void* dk;
    void* dv;
    long long n4;     // elements / 4 per tensor
    int gsplit;
};

template <class T>
__global__ void __launch_bounds__(256) fa_bwd_reduce_kernel(const ReduceParams p) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= 2 * p.n4) return;
    const int which = i >= p.n4;  // 0: dK, 1: dV
    const long long e = which ? i - p.n4 : i;
    const f32x4_t* src = reinterpret_cast<const f32x4_t*>(p.part) + (long long)which * p.gsplit * p.n4 + e;
    f32x4_t acc = src[0];
    for (int s = 1; s < p.gsplit; ++s) {
        const f32x4_t x = src[(long long)s * p.n4];
        acc[0] += x[0]; acc[1] += x[1]; acc[2] += x[2]; acc[3] += x[3];
    }
    u32x2_t u;
    u[0] = T::pack2(acc[0], acc[1]);
    u[1] = T::pack2(acc[2], acc[3]);
    reinterpret_cast<u32x2_t*>(which ? p.dv : p.dk)[e] = u;
}

// Head split of the dK/dV kernel: a GQA/MQA problem has few (batch, kv-head) units, so the query heads
// of a group are spread over `gsplit` workgroups (a divisor of g) until the grid covers the chip.
constexpr int kTargetWorkgroups = 256;  // MI355X: 256 CUs, one 512-thread workgroup each
inline int dkdv_gsplit(int B, int Hq, int Hkv, int Sk, int causal) {
    const int g = Hq / Hkv;
    const int nkb = (Sk + kKvBlock - 1) / kKvBlock;
    const long long base = (long long)B * Hkv * (causal ? (nkb + 1) / 2 : nkb);
    int sp = 1;
    while (sp < g && base * sp < kTargetWorkgroups) {
        int next = sp + 1;
        while (next < g && g % next != 0) ++next;
        sp = next;
    }
    return sp;
}
inline uint64_t delta_bytes(int B, int Hq, int Sq) {
    return (((uint64_t)B * Hq * Sq * sizeof(float)) + 255) / 256 * 256;
}

// ---- the 5-matmul backward (round 5): delta pass -> dK/dV kernel that also spills its packed dS -> dQ = dS K (fa_bwd_dqs_gfx950.hip).
// MEASURED (profiles/r5_bwd_spill.txt): it removes the recomputation (7 -> 5 tile matmuls) but moves 2 x 2 KB per (32 x 32) tile through
// the memory fabric -- 128 FLOP saved per byte moved, below the chip's ridge of ~315 -- and the dK/dV kernel, whose Q / dO re-reads
// already miss the 4 MB L2 half of the time, stalls on the added write stream (+18 .. +34 % cycles even on zero inputs); the dQ kernel
// then runs at the fabric's ~4.2 TB/s.  C2 1794 vs 1751 us, C3 472 .. 486 vs 474 us, D = 64 832 vs 668 us against the recompute pair.
// So it is NOT the mode of the large shapes.  Where the touched dS stays inside the 256 MB Infinity Cache (which keeps what a kernel wrote
// for the next one: tools/probe_mall.hip) it wins: B1 H32 S2048 (141 MB) 157.8 -> 129.7 us, B2 H32 S1024 106.6 -> 83.8, D = 64 B1 H16 S2048
// 99.5 -> 78.8; B1 H32 S4096 (537 MB) 432 -> 452.  Default ("auto"): the 5-matmul backward for problems whose touched dS is at most
// AULE_HIP_BWD_DS_AUTO_MB (160) and whose dK/dV grid takes the one-wave-per-SIMD kernel anyway, the recompute pair for everything else;
// AULE_HIP_BWD_MODE=spill | recompute pin either one.
// AULE_HIP_BWD_DS_CAP_MB (default 8192) bounds the workspace: the batch runs in chunks of as many elements as the caller's buffer holds.
// What the most recent backward launch of this process ran (aule_hip_debug_last_backward_route; tests pin the mode a shape takes with it)
std::atomic<int> g_last_bwd_route{0};
enum { kRouteSpill = 1, kRouteDq4 = 2, kRouteDkv4 = 4, kRouteDqOld = 8, kRouteDkvOld = 16, kRouteF32 = 32, kRouteDkv4K2 = 64 };

inline int bwd_mode() {   // 0: auto (by AULE_HIP_BWD_DS_AUTO_MB), 1: recompute, 2: spill wherever applicable
    static const int m = [] {
        const char* e = std::getenv("AULE_HIP_BWD_MODE");
        if (e == nullptr) return 0;
        return e[0] == 'r' ? 1 : (e[0] == 's' ? 2 : 0);
    }();
    return m;
}
inline uint64_t bwd_ds_auto_bytes() {
    static const uint64_t c = [] {
        const char* e = std::getenv("AULE_HIP_BWD_DS_AUTO_MB");
        const long long mb = e != nullptr ? std::atoll(e) : 160;
        return (uint64_t)(mb > 0 ? mb : 0) << 20;
    }();
    return c;
}
inline uint64_t bwd_ds_cap_bytes() {
    static const uint64_t c = [] {
        const char* e = std::getenv("AULE_HIP_BWD_DS_CAP_MB");
        const long long mb = e != nullptr ? std::atoll(e) : 8192;
        return (uint64_t)(mb > 0 ? mb : 0) << 20;
    }();
    return c;
}
// does the dispatcher take the one-wave-per-SIMD dK/dV kernel for these sizes?  (the grid rule of launch_bwd_16 below)
inline bool dkv4_by_grid(int B, int Hq, int Hkv, int Sk, int causal) {
    if (bwd_dkv4_forced()) return true;
    const int nkb = (Sk + kKvBlock - 1) / kKvBlock;
    const long long here = (long long)B * Hkv * (causal ? (nkb + 1) / 2 : nkb) * dkdv_gsplit(B, Hq, Hkv, Sk, causal);
    const int nkb4 = (Sk + 127) / 128;
    const long long there = (long long)B * Hkv * (causal ? (nkb4 + 1) / 2 : nkb4);
    return there >= 192 || there >= here;
}
// bytes of dS workspace per batch element if the sizes alone allow the 5-matmul backward, else 0
inline uint64_t spill_bytes_per_batch(int B, int Hq, int Hkv, int Sq, int Sk, int D, int causal, int dtype) {
    if (bwd_mode() == 1 || (dtype != kBF16 && dtype != kF16) || (D != 128 && D != 64)) return 0;
    if (Hkv <= 0 || Hq % Hkv != 0 || Sq <= 0 || Sk <= 0) return 0;
    BwdArgs t{};
    t.B = B; t.Hq = Hq; t.Hkv = Hkv; t.Sq = Sq; t.Sk = Sk; t.D = D; t.causal = causal; t.dtype = dtype; t.window = -1; t.coff = 0;
    if (!bwd_dkv4_applicable(t) || !bwd_dqs_applicable(t) || !dkv4_by_grid(B, Hq, Hkv, Sk, causal)) return 0;
    const uint64_t pb = (uint64_t)Hkv * (uint64_t)DsLayout::of(Hq, Hkv, Sq, Sk).group_bytes;
    if (pb > bwd_ds_cap_bytes()) return 0;
    if (bwd_mode() == 0) {   // auto: only problems whose TOUCHED dS (the causal half) fits the budget.  (The columns are allocated as full
        // squares -- the address is the stream position -- so a causal problem asks for up to twice the budget of workspace: INTEGRATION.md,
        // include/aule.h state it; only the touched half travels through the cache.)
        const uint64_t touched = (uint64_t)B * pb / (causal ? 2 : 1);
        if (touched > bwd_ds_auto_bytes()) return 0;
    }
    return pb;
}
inline uint64_t bwd_base_bytes(int B, int Hq, int Hkv, int Sq, int Sk, int D, int causal, int dtype, int device = -1) {
    uint64_t bytes = delta_bytes(B, Hq, Sq);
    if (dtype == kF32) bytes += aule_hip::bwd_f32_partial_bytes(B, Hq, Hkv, Sq, Sk, D, causal, device);   // small grids: the key / query range pieces' planes
    if (dtype != kF32) {
        bytes += 2 * delta_bytes(B, Hq, Sq);   // L' = LSE log2(e) and - delta, published with delta
        const int sp = dkdv_gsplit(B, Hq, Hkv, Sk, causal);
        if (sp > 1) bytes += 2ull * sp * B * Hkv * Sk * D * sizeof(float);
        bytes = (bytes + 255) / 256 * 256;
    }
    return bytes;
}

template <class T, int D>
int launch_bwd_16(const BwdArgs& a, hipStream_t stream) {
    // (delta = rowsum(O * dO) is computed inside the dQ kernel)
    BwdParams p;
    p.q = a.q; p.k = a.k; p.v = a.v; p.dout = a.dout; p.lse = a.lse; p.delta = a.delta;
    p.o = a.o; p.delta_out = a.delta;
    p.lse2_out = reinterpret_cast<float*>(reinterpret_cast<char*>(a.delta) + delta_bytes(a.B, a.Hq, a.Sq));
    p.ndelta_out = reinterpret_cast<float*>(reinterpret_cast<char*>(a.delta) + 2 * delta_bytes(a.B, a.Hq, a.Sq));
    p.dq = a.dq; p.dk = a.dk; p.dv = a.dv;
    p.B = a.B; p.Hq = a.Hq; p.Hkv = a.Hkv; p.Sq = a.Sq; p.Sk = a.Sk;
    p.c = a.scale * kLog2e;
    p.scale = a.scale;
    p.window = a.window > 0 ? a.window : 0;
    p.coff = a.causal ? a.coff : 0;
#ifdef AULE_DEBUG_HOOKS
    // debug library only: AULE_DBG_BWD_ONLY=dq / =dkv launches one of the two kernels (per-kernel times from tools/cbench.cpp
    // without a profiler; the workspace keeps delta / L' of an earlier full call)
    static const int only = [] {
        const char* e = std::getenv("AULE_DBG_BWD_ONLY");
        if (e == nullptr) return 0;
        return std::strcmp(e, "dq") == 0 ? 1 : (std::strcmp(e, "dkv") == 0 ? 2 : 0);   // anything else: both kernels
    }();
#else
    constexpr int only = 0;
#endif
    // ---- the 5-matmul backward: no recomputation (see spill_bytes_per_batch above; fa_bwd_dqs_gfx950.hip)
    if constexpr (D == 128 || D == 64) {
        const uint64_t pb = (only == 0 && a.dbg == nullptr && a.dbg_dq == nullptr && !dkv4_timeline_wanted() && a.window <= 0 && bwd_dkv4_applicable(a) &&
                             bwd_dqs_applicable(a))
                                ? spill_bytes_per_batch(a.B, a.Hq, a.Hkv, a.Sq, a.Sk, D, a.causal, a.dtype) : 0;
        const uint64_t base = bwd_base_bytes(a.B, a.Hq, a.Hkv, a.Sq, a.Sk, D, a.causal, a.dtype);
        const uint64_t nb = (pb > 0 && a.ws_bytes > base) ? (a.ws_bytes - base) / pb : 0;
        if (nb >= 1) {
            g_last_bwd_route = kRouteSpill | kRouteDkv4;
            int rc = launch_bwd_delta16(a, p.lse2_out, p.ndelta_out, stream);
            if (rc) return rc;
            const size_t rq = (size_t)a.Hq * a.Sq, rk = (size_t)a.Hkv * a.Sk;   // rows per batch element
            for (int b0 = 0; b0 < a.B; b0 += (int)nb) {
                BwdArgs c = a;
                c.B = a.B - b0 < (int)nb ? a.B - b0 : (int)nb;
                c.q = reinterpret_cast<const char*>(a.q) + b0 * rq * D * 2; c.dout = reinterpret_cast<const char*>(a.dout) + b0 * rq * D * 2;
                c.k = reinterpret_cast<const char*>(a.k) + b0 * rk * D * 2; c.v = reinterpret_cast<const char*>(a.v) + b0 * rk * D * 2;
                c.dq = reinterpret_cast<char*>(a.dq) + b0 * rq * D * 2;
                c.dk = reinterpret_cast<char*>(a.dk) + b0 * rk * D * 2; c.dv = reinterpret_cast<char*>(a.dv) + b0 * rk * D * 2;
                c.lse2 = p.lse2_out + b0 * rq; c.ndelta = p.ndelta_out + b0 * rq;
                c.ds = reinterpret_cast<char*>(a.delta) + base;
                rc = launch_bwd_dkv4(c, stream);
                if (rc) return rc;
                rc = launch_bwd_dqs(c, stream);
                if (rc) return rc;
            }
            return 0;
        }
    }
    // The one-wave-per-SIMD dQ kernel (fa_bwd_dq4_gfx950.hip: 64 query rows per wave, every K / V fragment read feeds two row
    // blocks; dQ bit-identical to this file's kernel) wherever it can run and the grid has at least 128 work items: ahead or level
    // on all ten shapes of tools/cb_rule_dq.sh (whole backward -0.2 .. -2.5 %, the 128-item grids level), behind on grids of a few
    // workgroups (its three-stage stream start and 64-row prologue).  AULE_HIP_BWD_DQ=old|new pin either one.
    const auto dq4_items = [&] {
        const int nqb = (a.Sq + kDqQBlock - 1) / kDqQBlock;
        return (long long)a.B * a.Hq * (a.causal ? (nqb + 1) / 2 : nqb);
    };
    // D = 64 (round 4): ahead on small grids too (16 .. 64 work items: +1.9 .. +7.5 % on the whole backward, profiles/r4_bwd_d64_dkv4.txt) -- no grid rule there.
    const bool use_dq4 = (D == 128 || D == 64) && a.dbg_dq == nullptr && bwd_dq4_applicable(a) && (bwd_dq4_mode() == 2 || D == 64 || dq4_items() >= 128);
    g_last_bwd_route = 0;
    if (only != 2 && use_dq4) {
        g_last_bwd_route |= kRouteDq4;
        int rc = launch_bwd_dq4(a, p.lse2_out, p.ndelta_out, stream);
        if (rc) return rc;
    } else if (only != 2) {
        g_last_bwd_route |= kRouteDqOld;
        const int nqb = (a.Sq + kDqQBlock - 1) / kDqQBlock;
        p.nblk = a.causal ? (nqb + 1) / 2 : nqb;  // causal: one workgroup per Q-block pair (i, n-1-i)
        p.gsplit = 1;
        p.part = nullptr;
        const dim3 grid((unsigned)(p.nblk * a.B * a.Hq)), block(512);
        p.dbg = a.dbg_dq;
        bool tl_done = false;
#ifdef AULE_DEBUG_HOOKS
        if constexpr (std::is_same<T, Bf16Traits>::value && D == 128) {
            if (a.dbg_dq != nullptr && a.causal) {   // timeline build (tools/timeline_bwd.py dq): bf16 D128 causal only
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&fa_bwd_dq_kernel<T, D, true, true>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, DqCfg<D>::LDS);
                hipLaunchKernelGGL((fa_bwd_dq_kernel<T, D, true, true>), grid, block, DqCfg<D>::LDS, stream, p);
                tl_done = true;
            }
        }
#endif
        if (tl_done) {
        } else if (a.causal)
            hipLaunchKernelGGL((fa_bwd_dq_kernel<T, D, true>), grid, block, DqCfg<D>::LDS, stream, p);
        else
            hipLaunchKernelGGL((fa_bwd_dq_kernel<T, D, false>), grid, block, DqCfg<D>::LDS, stream, p);
        int rc = (int)hipGetLastError();
        if (rc) return rc;
    }
    if (only == 1) return 0;
    // The one-wave-per-SIMD dK/dV kernel (fa_bwd_dkv4_gfx950.hip: 128-key blocks, the whole GQA group inside a workgroup, no head
    // split / partials / reduce) wherever it covers the chip (>= 192 work items) or has at least as many work items as this
    // file's kernel would (256-key blocks x head split).  Same box, whole backward, tools/cb_rule.sh: ahead on every shape of the
    // spread (MHA / GQA, causal or not, ragged, fp16: -0.5 .. -24 %); what stays here is the tiny grid with a big group (fp16 MQA
    // 32/1 S8192: 32 work items there against 512 here).
    const auto use_dkv4 = [&] {
        if ((D != 128 && D != 64) || !(a.dbg == nullptr || dkv4_timeline_wanted()) || !bwd_dkv4_applicable(a)) return false;   // (timeline instances: bf16, D = 128 and, round 5, D = 64)
        if (dkv4_timeline_wanted()) return true;
        return dkv4_by_grid(a.B, a.Hq, a.Hkv, a.Sk, a.causal);   // (ONE statement of the grid rule: the auto mode's workspace plan asks the same helper)
    };
    if (use_dkv4())
    {
        g_last_bwd_route |= kRouteDkv4;
        BwdArgs b = a;
        if (bwd_dkv4_k2(b)) g_last_bwd_route |= kRouteDkv4K2;
        b.lse2 = p.lse2_out;
        b.ndelta = p.ndelta_out;
        return launch_bwd_dkv4(b, stream);
    }
    {
        g_last_bwd_route |= kRouteDkvOld;
        const int nkb = (a.Sk + kKvBlock - 1) / kKvBlock;
        p.nblk = a.causal ? (nkb + 1) / 2 : nkb;  // causal: one workgroup per block pair (i, n-1-i)
        p.gsplit = dkdv_gsplit(a.B, a.Hq, a.Hkv, a.Sk, a.causal);
        p.part = reinterpret_cast<float*>(reinterpret_cast<char*>(a.delta) + 3 * delta_bytes(a.B, a.Hq, a.Sq));
        const dim3 grid((unsigned)(p.nblk * a.B * a.Hkv * p.gsplit)), block(512);
        p.dbg = a.dbg;
        bool tl_done = false;
#ifdef AULE_DEBUG_HOOKS
        if constexpr (std::is_same<T, Bf16Traits>::value && D == 128) {
            if (a.dbg != nullptr && a.causal) {   // timeline build (tools/timeline_bwd.py): bf16 D128 causal only
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&fa_bwd_dkdv_kernel<T, D, true, true>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, DkvCfg<D>::LDS);
                hipLaunchKernelGGL((fa_bwd_dkdv_kernel<T, D, true, true>), grid, block, DkvCfg<D>::LDS, stream, p);
                tl_done = true;
            }
        }
#endif
        if (tl_done) {
        } else if (a.causal)
            hipLaunchKernelGGL((fa_bwd_dkdv_kernel<T, D, true>), grid, block, DkvCfg<D>::LDS, stream, p);
        else
            hipLaunchKernelGGL((fa_bwd_dkdv_kernel<T, D, false>), grid, block, DkvCfg<D>::LDS, stream, p);
        int rc = (int)hipGetLastError();
        if (rc || p.gsplit == 1) return rc;
        ReduceParams r;
        r.part = p.part; r.dk = a.dk; r.dv = a.dv; r.gsplit = p.gsplit;
        r.n4 = (long long)a.B * a.Hkv * a.Sk * D / 4;
        hipLaunchKernelGGL((fa_bwd_reduce_kernel<T>), dim3((unsigned)((2 * r.n4 + 255) / 256)), dim3(256), 0, stream, r);
        return (int)hipGetLastError();
    }
}

template <class T, int D>
int set_attr_bwd() {
    int rc = 0;
    rc |= (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&fa_bwd_dq_kernel<T, D, true>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, DqCfg<D>::LDS);
    rc |= (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&fa_bwd_dq_kernel<T, D, false>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, DqCfg<D>::LDS);
    rc |= (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&fa_bwd_dkdv_kernel<T, D, true>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, DkvCfg<D>::LDS);
    rc |= (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&fa_bwd_dkdv_kernel<T, D, false>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, DkvCfg<D>::LDS);
    return rc;
}

}  // namespace

int launch_bwd_f32(const BwdArgs& a, hipStream_t stream);  // fa_bwd_f32.hip
int configure_bwd_f32();

// Shared with fa_bwd_f32.hip
int launch_delta_f32(const BwdArgs& a, hipStream_t stream) {
    DeltaParams dp;
    dp.o = a.o; dp.dout = a.dout; dp.delta = a.delta;
    dp.rows = (long long)a.B * a.Hq * a.Sq;
    const dim3 block(256);
    if (a.D == 128) {
        const dim3 grid((unsigned)((dp.rows + 7) / 8));
        hipLaunchKernelGGL((fa_bwd_delta_f32_kernel<128>), grid, block, 0, stream, dp);
    } else if (a.D == 64) {
        const dim3 grid((unsigned)((dp.rows + 15) / 16));
        hipLaunchKernelGGL((fa_bwd_delta_f32_kernel<64>), grid, block, 0, stream, dp);
    } else if (a.D == 32) {
        const dim3 grid((unsigned)((dp.rows + 31) / 32));
        hipLaunchKernelGGL((fa_bwd_delta_f32_kernel<32>), grid, block, 0, stream, dp);
    } else {
        return -1;
    }
    return (int)hipGetLastError();
}

// What launch_bwd NEEDS (delta, L', - delta, the head-split partials) ...
uint64_t bwd_workspace_min_bytes(int B, int Hq, int Hkv, int Sq, int Sk, int D, int causal, int dtype, int device) {
    return bwd_base_bytes(B, Hq, Hkv, Sq, Sk, D, causal, dtype, device);
}
// ... and what it WANTS: plus the dS workspace of the 5-matmul backward for as many batch elements as fit the cap (at least one).
// A caller that passes only the minimum gets the recompute pair.
// (windowed: the dispatcher keeps windowed problems on the recompute pair -- no dS room is asked for: ADVICE r5)
uint64_t bwd_workspace_bytes(int B, int Hq, int Hkv, int Sq, int Sk, int D, int causal, int dtype, int device, bool windowed) {
    uint64_t bytes = bwd_base_bytes(B, Hq, Hkv, Sq, Sk, D, causal, dtype, device);
    const uint64_t pb = windowed ? 0 : spill_bytes_per_batch(B, Hq, Hkv, Sq, Sk, D, causal, dtype);
    if (pb > 0) {
        uint64_t nb = bwd_ds_cap_bytes() / pb;
        if (nb > (uint64_t)B) nb = (uint64_t)B;
        bytes += nb * pb;
    }
    return bytes;
}

int bwd_last_route() { return g_last_bwd_route.load(); }

int launch_bwd(const BwdArgs& a, hipStream_t stream) {
    if (a.dtype == kF32) { g_last_bwd_route = kRouteF32; return launch_bwd_f32(a, stream); }
    if (a.dtype == kBF16) {
        if (a.D == 128) return launch_bwd_16<Bf16Traits, 128>(a, stream);
        if (a.D == 64) return launch_bwd_16<Bf16Traits, 64>(a, stream);
        if (a.D == 32) return launch_bwd_16<Bf16Traits, 32>(a, stream);
    } else if (a.dtype == kF16) {
        if (a.D == 128) return launch_bwd_16<F16Traits, 128>(a, stream);
        if (a.D == 64) return launch_bwd_16<F16Traits, 64>(a, stream);
        if (a.D == 32) return launch_bwd_16<F16Traits, 32>(a, stream);
    }
    return -1;
}

int configure_bwd() {
    int rc = 0;
    rc |= set_attr_bwd<Bf16Traits, 128>();
    rc |= set_attr_bwd<Bf16Traits, 64>();
    rc |= set_attr_bwd<Bf16Traits, 32>();
    rc |= set_attr_bwd<F16Traits, 128>();
    rc |= set_attr_bwd<F16Traits, 64>();
    rc |= set_attr_bwd<F16Traits, 32>();
    rc |= configure_bwd_f32();
    rc |= configure_bwd_dkv4();
    rc |= configure_bwd_dq4();
    rc |= configure_bwd_dqs();
    return rc;
}

}  // namespace aule_hip
