// fa_kernels.h -- host-visible launch interface of the gfx950 attention kernels.
//
// Plain C++ (no torch, no HIP types beyond hipStream_t) so that aule_capi.cpp
// can call the launchers.  Every launcher is asynchronous on `stream` and
// returns a hipError_t-compatible int (0 = success).
//
// Tensor layout (row-major contiguous, as the reference's Triton path makes
// them: python/aule/triton_flash_amd.py:404-407):
//   Q, O, dO, dQ : [B, Hq,  Sq, D]      K, V, dK, dV : [B, Hkv, Sk, D]
//   LSE, delta   : [B, Hq, Sq] fp32
#pragma once
#include <hip/hip_runtime_api.h>
#include <stdint.h>

#include <atomic>

namespace aule_hip {

enum DType : int { kF32 = 0, kF16 = 1, kBF16 = 2 };

struct FwdArgs {
    const void* q;
    const void* k;
    const void* v;
    void* o;
    float* lse;  // may be null
    int B, Hq, Hkv, Sq, Sk, D;
    float scale;  // softmax scale (already defaulted by the caller)
    int causal;
    int dtype;
    int window = -1;  // sliding window: key j visible to query i only if i - j < window (<= 0: off)
    int coff = 0;     // causal position offset: query i sits at position i + coff (0 = the reference's top-left
                      // rule; Sk - Sq = bottom-right alignment); also shifts the window
    // Partials of the two-launch short-query paths: the caller's buffer when it is large enough (no allocation at
    // all -- what a hipGraph capture wants: hipMallocAsync / hipFreeAsync become graph nodes that cost more than the
    // kernels of a decode step), otherwise a stream-ordered allocation.
    void* ws = nullptr;
    uint64_t ws_bytes = 0;
    uint64_t* query_ws = nullptr;   // dry run: the launcher stores the bytes it would need and launches nothing
    // Fused query rotation (fwd_rope_fusable() shapes only): Q is rotated on its way into the kernel's registers with the
    // half-split pairs of rope_gfx950.hip; K must arrive rotated.  Tables [rope_rows, rope_pitch] fp32, query i -> row i + rope_pos.
    const float* rope_cos = nullptr;
    const float* rope_sin = nullptr;
    int rope_rows = 0, rope_pitch = 0, rope_pos = 0;
    int device = -1;   // the descriptor's device ordinal (-1: the current device): the grid-sizing queries ask THAT device's CU
                       // count, also from the entry points that never switch devices (workspace size, route, fusable)
};

// Compute units of `device` (-1: the current one), asked once per device id and process: launch paths call this several times
// per launch (route, plan, grid).
inline int device_cu_count(int device) {
    static std::atomic<int> cached[64];   // (zero-initialised; API threads may race to fill an entry with the same value)
    int dev = device;
    if (dev < 0 && hipGetDevice(&dev) != hipSuccess) return 256;
    if (dev < 0 || dev >= 64) return 256;
    const int have = cached[dev].load(std::memory_order_relaxed);
    if (have > 0) return have;
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) return 256;
    cached[dev].store(n, std::memory_order_relaxed);
    return n;
}

// The workspace of one launch: the caller's buffer, or hipMallocAsync / hipFreeAsync on the launch stream.
struct ScopedWorkspace {
    void* ptr = nullptr;
    bool owned = false;
    hipStream_t stream;
    hipError_t err = hipSuccess;
    ScopedWorkspace(size_t bytes, void* user, uint64_t user_bytes, hipStream_t s) : stream(s) {
        if (user != nullptr && user_bytes >= bytes && (reinterpret_cast<uintptr_t>(user) & 15) == 0) {
            ptr = user;
            return;
        }
        err = hipMallocAsync(&ptr, bytes, s);
        owned = err == hipSuccess;
    }
    ~ScopedWorkspace() {
        if (owned) (void)hipFreeAsync(ptr, stream);
    }
    ScopedWorkspace(const ScopedWorkspace&) = delete;
    ScopedWorkspace& operator=(const ScopedWorkspace&) = delete;
};

struct BwdArgs {
    const void* q;
    const void* k;
    const void* v;
    const void* o;
    const void* dout;
    const float* lse;
    void* dq;
    void* dk;
    void* dv;
    float* delta;  // workspace of bwd_workspace_bytes(): delta [B,Hq,Sq] fp32 first
    const float* lse2 = nullptr;   // internal (16-bit path): L' = LSE log2(e) [B,Hq,Sq], published by the dQ kernel behind delta
    const float* ndelta = nullptr; // internal (16-bit path): - delta, behind L' (the C operand the one-wave-per-SIMD dK/dV kernel starts dP from)
    int B, Hq, Hkv, Sq, Sk, D;
    float scale;
    int causal;
    int dtype;
    int window = -1;  // as FwdArgs::window (the reference's backward ignores it; this one honours it)
    int coff = 0;     // as FwdArgs::coff
    unsigned long long* dbg = nullptr;   // debug: s_memtime stamps of the dK/dV kernel's workgroup 0 (bf16 D128 causal)
    unsigned long long* dbg_dq = nullptr;   // ... of the dQ kernel's workgroup 0
    void* ds = nullptr;            // internal (16-bit path, 5-matmul backward): the dS workspace of this call's batch chunk (DsLayout)
    uint64_t ws_bytes = 0;         // bytes behind `delta` (0 = just bwd_workspace_min_bytes(): the recompute pair runs)
    int device = -1;               // the descriptor's device ordinal (-1: the current one): the grid-sizing rules ask THAT device's CU count (as FwdArgs::device)
};

// The dS workspace of the 5-matmul backward (round 5; fa_bwd_dkv4_gfx950.hip SPILL instances write it, fa_bwd_dqs_gfx950.hip reads it).
// A UNIT is the packed 16-bit dS of one (32-key block, 32-row query block) tile exactly as the dK/dV kernel holds it for its own
// dK MFMAs: 2 KB = [kk = 16-row query step][lane = key n + 32 hi][16 bytes = query rows 16 kk + 4 hi + {0..3}, 16 kk + 8 + 4 hi + {0..3}].
// Units of one (batch, KV head) GROUP and one 32-key block kb32 form a COLUMN of xs = g * nq32 units in the dK/dV kernel's stream
// order -- block-major, head-minor since round 6: query block qb32 of head hh of the group -> x = g * qb32 + hh -- so that the
// writer's address is its loop counter (its stream starts at the first query block fq that sees the 128-key block: x0 = g * fq).
// Columns are padded to whole 128-key blocks (nkb32p = 4 * ceil(Sk / 128)): every wave of the dK/dV kernel owns a column.
struct DsLayout {
    int nq32, nkb32p;
    long long xs;              // units per column
    long long group_bytes;     // nkb32p * xs * 2048
    static DsLayout of(int Hq, int Hkv, int Sq, int Sk) {
        DsLayout l;
        l.nq32 = (Sq + 31) / 32;
        l.nkb32p = 4 * ((Sk + 127) / 128);
        l.xs = (long long)(Hq / (Hkv > 0 ? Hkv : 1)) * l.nq32;
        l.group_bytes = (long long)l.nkb32p * l.xs * 2048;
        return l;
    }
};

// Paged-KV decode (python/aule/triton_flash_amd.py:543-737): one query token per sequence.
//   q, out : [B, Hq, D]      k_cache, v_cache : [num_blocks, block_size, Hkv, D]   (16-bit dtypes)
//   block_tables : [B, max_blocks] int32 (physical block of each logical block), context_lens : [B] int32
struct PagedArgs {
    const void* q;
    const void* k_cache;
    const void* v_cache;
    void* out;
    const int* block_tables;
    const int* context_lens;
    int B, Hq, Hkv, D;
    int block_size, max_blocks;
    float scale;
    int window;   // > 0: attend only to the last `window` positions (context_len - 1 - pos < window)
    int dtype;
    void* ws = nullptr;             // as FwdArgs::ws / ws_bytes / query_ws
    uint64_t ws_bytes = 0;
    uint64_t* query_ws = nullptr;
};

// Rotary embedding pass (rope_gfx950.hip): x [nheads, S, D] with `pitch` elements per row, tables [>= S + pos_offset, D/2]
// fp32; layout 0 = half-split pairs (p, p + D/2), 1 = interleaved pairs (2p, 2p + 1); in == out is allowed.
struct RopeArgs {
    const void* in;
    void* out;
    const float* cos;
    const float* sin;
    long long nheads;   // B * H
    int S, D, pitch;
    int layout, inverse, pos_offset;
    int dtype;
    int table_pitch = 0;   // floats per table row; 0 = D/2
};
int launch_rope(const RopeArgs& a, hipStream_t stream);

// Returns 0 on success, a hipError_t value on launch failure, -1 for an
// unsupported (dtype, D) combination.
int launch_paged_decode(const PagedArgs& a, hipStream_t stream);
int launch_fwd(const FwdArgs& a, hipStream_t stream);
// merge partials [npart][B*Hkv*nrt*32][D+2] fp32 (un-normalised O, m in log2 units, l) into O / LSE (fa_fwd_splitkv_gfx950.hip)
int launch_splitkv_combine(const FwdArgs& a, float* part, int npart, int nrt, hipStream_t stream);
int fwd_route(const FwdArgs& a);
// launch_fwd honours FwdArgs::rope_* for these arguments (otherwise it refuses them: rotate Q with launch_rope first)
bool fwd_rope_fusable(const FwdArgs& a);
// the split plan of route 7 as integers (tests): fwd_split_plan_dump in fa_fwd_w4_gfx950.hip, the plan itself in fa_fwd_split.h
int fwd_split_plan_dump(const FwdArgs& a, int* out, int cap);
// blockIdx -> (batch, kv head, q head, block) of decode_work (ranked = 0) / decode_work_ranked (1) on the host (tests): fa_fwd_f32.hip
void work_order_dump(int ranked, int bid, int B, int Hq, int Hkv, int nblk, int flag, int* out4);
// bytes of workspace launch_fwd / launch_paged_decode would allocate for these arguments (0: single-launch path)
uint64_t fwd_workspace_bytes(FwdArgs a);
uint64_t paged_workspace_bytes(PagedArgs a);   // 0 fp32, 1 ping-pong, 2 in-wave, 3 v1, 4 split-KV, 5 tiled + packed rows + KV splits (host logic only)
int launch_bwd(const BwdArgs& a, hipStream_t stream);
// bit mask of what the most recent launch_bwd of this process ran: 1 the 5-matmul mode (delta pass + spilling dK/dV kernel + dQ = dS K),
// 2 / 4 the one-wave-per-SIMD dQ / dK/dV kernel, 8 / 16 their two-waves-per-SIMD predecessors, 32 the fp32 kernels, 64 (with 4) the D = 64
// dK/dV instance with two key blocks per wave; 0 before the first
int bwd_last_route();

// Bytes of device workspace launch_bwd needs: delta [B,Hq,Sq] fp32, plus (16-bit GQA/MQA problems that
// do not fill the chip) fp32 dK/dV partials of the head-split dK/dV kernel.
uint64_t bwd_workspace_bytes(int B, int Hq, int Hkv, int Sq, int Sk, int D, int causal, int dtype, int device = -1, bool windowed = false);
// ... of which launch_bwd cannot do without (the rest is the dS workspace of the 5-matmul backward: fa_bwd_gfx950.hip)
uint64_t bwd_workspace_min_bytes(int B, int Hq, int Hkv, int Sq, int Sk, int D, int causal, int dtype, int device = -1);

// Set the max-dynamic-LDS attribute on every kernel (call once per device).
int configure_kernels();

}  // namespace aule_hip
