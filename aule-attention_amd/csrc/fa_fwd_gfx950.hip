// fa_fwd_gfx950.hip -- forward dispatch for MI355X (gfx950 / CDNA4).
//
// Replaces, behind aule_attention_forward_ex, the Triton launch of the reference
// (python/aule/triton_flash_amd.py:393-445 FlashAttentionAMDFunc.forward and its
// kernel :97-240) and the HIP stub slot (src/backends/attention_hip.cpp:22-119).
// Numerics contract: s = scale * q.k ; top-left causal
// mask (j <= i) ; online softmax with fp32 m, l, acc ; P cast to the V dtype
// before the PV product (triton_flash_amd.py:222) ; LSE = m + ln(l) (:237).
//
// The kernels live in their own files; this one only picks among them (host logic):
//   fa_fwd_w4_gfx950.hip       16-bit, one wave per SIMD (4 x 64 query rows), persistent part lists: the default for tiled problems;
//                              small grids as key-range pieces + merge (fa_fwd_split.h)
//   fa_fwd_pp_gfx950.hip       16-bit, two waves per SIMD, one workgroup per Q-block pair: sliding windows below 128 keys or without the causal rule, fewer
//                              than four KV tiles per Q block, D = 32, scale = 0, and the SPLIT instances (packed rows + KV splits) for short queries
//   fa_fwd_splitkv_gfx950.hip  16-bit, wave-per-chunk split-KV (decode streaming corner) and paged decode
//   fa_fwd_f32.hip             fp32 I/O
// Common to the 16-bit kernels: "swapped" S^T = K.Q^T so that a lane owns one query row; P stays in registers as
// the B operand of O^T += V^T.P^T; K row-major and V sub-tiled in LDS; all blocks of one (batch, kv-head) on one XCD.
// (Retired, in the history: the lock-step kernel of round 1 and the in-wave pipelined variants -- fa_fwd_iw_gfx950.hip -- in
// round 2; the two-waves-per-SIMD persistent tile stream -- fa_fwd_ps_gfx950.hip, routes 6 and 7 of rounds 2-3 -- in round 4,
// when the one-wave-per-SIMD kernel took its small-grid split and the ping-pong kernel its remaining shapes.)
#include <cstdio>
#include <cstdlib>

#include "fa_device.h"
#include "fa_kernels.h"

namespace aule_hip {
int launch_fwd_f32(const FwdArgs& a, hipStream_t stream);  // fa_fwd_f32.hip
int configure_fwd_f32();
int launch_fwd_pp(const FwdArgs& a, hipStream_t stream);   // fa_fwd_pp_gfx950.hip
int launch_fwd_pp_split(const FwdArgs& a, hipStream_t stream);
bool pp_split_applicable(const FwdArgs& a);
int configure_fwd_pp();
int launch_fwd_w4(const FwdArgs& a, hipStream_t stream);   // fa_fwd_w4_gfx950.hip (one wave per SIMD, 4 x 64 rows)
bool fwd_w4_applicable(const FwdArgs& a);
bool fwd_w4_split_applicable(const FwdArgs& a);            // small grids: pairs cut into key ranges, partials + merge
int launch_fwd_w4_split(const FwdArgs& a, hipStream_t stream);
int configure_fwd_w4();

// AULE_HIP_FWD_KERNEL=pp keeps every tiled problem on the ping-pong kernel (A/B measurements against the one-wave-per-SIMD kernel)
static int fwd_kernel_choice() {
    static const int v = [] {
        const char* e = getenv("AULE_HIP_FWD_KERNEL");
        if (e == nullptr || e[0] == 0) return 0;
        if (e[0] == 'p' && e[1] == 'p' && e[2] == 0) return 4;
        if (e[0] == 'w' && e[1] == '4' && e[2] == 0) return 0;
        // ("ps" named the persistent tile stream retired in round 4 and used to be ignored silently: an A/B run then
        // measured the default kernel twice)
        fprintf(stderr, "libaule: WARNING: AULE_HIP_FWD_KERNEL=%s is not a kernel of this build (known: pp, w4) -- the default dispatch runs\n", e);
        return 0;
    }();
    return v;
}
// AULE_HIP_FWD_SOFTMAX=classic asks for the online softmax throughout: the one-wave-per-SIMD kernel has no online form (its
// fall-back is a second pass with the exact row maximum), so such runs stay on the ping-pong kernel
static bool softmax_classic() {
    static const int v = [] {
        const char* e = getenv("AULE_HIP_FWD_SOFTMAX");
        return (e != nullptr && e[0] == 'c') ? 1 : 0;
    }();
    return v == 1;
}
static bool use_w4(const FwdArgs& a) { return fwd_kernel_choice() == 0 && !softmax_classic() && fwd_w4_applicable(a); }
static bool use_w4_split(const FwdArgs& a) { return fwd_kernel_choice() == 0 && !softmax_classic() && fwd_w4_split_applicable(a); }

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
    const bool tiled_ok = pp_split_applicable(a);
    if (wave_ok && tiled_ok) {
        const long long units = (long long)a.B * a.Hkv;
        const long long rows = (long long)(a.Hq / a.Hkv) * a.Sq;
        const double kv_bytes = 2.0 * (double)units * a.Sk * a.D * 2.0;
        return (units >= 32 && rows <= 16 && kv_bytes >= 100e6) ? 4 : 5;
    }
    return wave_ok ? 4 : (tiled_ok ? 5 : 0);
}

// Which kernel launch_fwd() picks for `a` (host logic only, no device work): 0 fp32, 1 ping-pong, 4 split-KV,
// 5 ping-pong kernel with packed rows + KV splits, 7 one-wave-per-SIMD kernel with every pair of causal Q blocks (every
// non-causal block) cut into key ranges (small grids; partials + merge), 8 one-wave-per-SIMD kernel (4 x 64 rows)
// (2, 3 and 6 were the removed in-wave, lock-step and two-waves-per-SIMD stream kernels).  Lets the tests pin the path a shape
// exercises.
int fwd_route(const FwdArgs& a) {
    if (a.dtype == kF32) return 0;
    const int sq = short_query_route(a);
    if (sq) return sq;
    if (use_w4_split(a)) return 7;
    return use_w4(a) ? 8 : 1;
}

// Which problems the forward rotates Q for by itself (half-split pairs, K already rotated): what the one-wave-per-SIMD kernel
// takes (its applicability rule looks at the table geometry too).
bool fwd_rope_fusable(const FwdArgs& a) { return fwd_route(a) == 8; }

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
    if (a.rope_cos != nullptr && !fwd_rope_fusable(a)) return -1;   // only the one-wave-per-SIMD kernel rotates Q itself
    const int sq = a.dtype == kF32 ? 0 : short_query_route(a);
    if (sq == 4) return launch_fwd_splitkv(a, stream);
    if (sq == 5) return launch_fwd_pp_split(a, stream);
    if (sq == 0 && a.dtype != kF32 && use_w4_split(a)) return launch_fwd_w4_split(a, stream);
    if (a.dtype == kF32) return launch_fwd_f32(a, stream);   // (small grids: key-range pieces + merge; answers the dry run itself)
    if (a.query_ws != nullptr) return 0;   // single-launch paths need no workspace
    if (use_w4(a)) return launch_fwd_w4(a, stream);
    return launch_fwd_pp(a, stream);   // short / non-causal windows, fewer than four KV tiles per Q block, D = 32, scale = 0, AULE_HIP_FWD_KERNEL=pp
}

int configure_fwd() {
    int rc = 0;
    rc |= configure_fwd_f32();
    rc |= configure_fwd_pp();
    rc |= configure_fwd_w4();
    return rc;
}

}  // namespace aule_hip
