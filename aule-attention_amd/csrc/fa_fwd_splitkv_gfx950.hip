// fa_fwd_splitkv_gfx950.hip -- forward for SHORT query sequences against long K/V (decode-like cross-attention,
// SURVEY.md 8d points C5b / C5c): the regime where the tiled kernels cannot fill the chip -- a 256-row Q block per
// workgroup leaves B*Hq*ceil(Sq/256) workgroups, 32 for C5b -- and the bound is HBM (K and V are read once,
// arithmetic intensity ~32 FLOP/B at Sq = 1).
//
//   * GQA/MQA packing: the Hq/Hkv query heads that share one K/V head are stacked into the ROW dimension
//     (row r of a (batch, kv-head) unit = (head r / Sq of the group, query r % Sq)), so MQA decode with 32 heads
//     is exactly one 32-row MFMA tile instead of 32 one-row problems, and K/V are streamed once per unit.
//   * split-KV: the key range is cut into chunks of `chunk_tiles` 32-key tiles, ONE WAVE per chunk (4 waves per
//     workgroup); every wave keeps its own online-softmax state and writes an un-normalised partial
//     (m, l, O) in fp32; fa_fwd_splitkv_combine merges the partials of a row (log-sum-exp merge), casts O and
//     writes LSE.  ~2048 waves are launched whatever the shape.
//   * per wave and 32-key tile: K fragments go from global memory straight into the MFMA A operand (row = key,
//     16 contiguous bytes per lane: no LDS), V goes through a wave-private LDS tile in the [kv/4][d/16][4][16]
//     sub-tile layout for ds_read_b64_tr_b16 (same layout and operand maps as fa_fwd_pp_gfx950.hip), S^T = K.Q^T
//     with a lane owning one packed row, O^T += V^T.P^T.
// Non-causal only (with the reference's top-left causal rule a short query sequence sees only its first Sq keys,
// which the tiled kernels handle).  Reference semantics: python/aule/triton_flash_amd.py:97-240 (same math,
// GQA head map :126-127).
#include "fa_device.h"
#include "fa_kernels.h"

namespace aule_hip {
namespace {

struct SplitParams {
    const void* q;
    const void* k;
    const void* v;
    void* o;
    float* lse;
    float* part;   // [npart][rows_total][D + 2] fp32: O (un-normalised), m (log2 units), l
    int B, Hq, Hkv, Sq, Sk;
    float c;       // |scale| * log2(e); the sign goes into Q
    int negq;
    int nrt;          // 32-row tiles per (batch, kv-head) unit
    int chunk_tiles;  // 32-key tiles per wave
    int npart;        // partials per row = 4 * gridDim.x
    int rows_total;   // B * Hkv * nrt * 32
    // paged KV cache (decode, Sq = 1; python/aule/triton_flash_amd.py:543-737): K/V = [num_blocks, block_size, Hkv, D]
    const int* block_tables;   // [B, max_blocks] physical block of each logical block
    const int* context_lens;   // [B] keys per sequence
    int block_size, max_blocks;
    int window;                // > 0: only the last `window` positions (context_len - 1 - pos < window)
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t skv_srd(const void* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}

template <class T, int D, bool PAGED>
__global__ void __launch_bounds__(256) fa_fwd_splitkv_kernel(const SplitParams p) {
    using v8 = typename T::v8;
    constexpr int RB = D * 2, KS = D / 16, DB = D / 32, CPR = RB / 16;
    constexpr int VT = 32 * RB;           // one wave's V tile
    constexpr int NV = (32 * CPR) / 64;   // 16-byte chunks per lane and tile
    __shared__ __attribute__((aligned(16))) char smem[4 * VT];

    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    char* const Vw = smem + wave * VT;

    const int g = p.Hq / p.Hkv;
    const int unit = blockIdx.y / p.nrt, rt = blockIdx.y % p.nrt;
    const int b = unit / p.Hkv, hk = unit % p.Hkv;
    const int Sq = p.Sq;
    // paged: keys of this sequence, read on the device and bounded by what the block table can address (a stale or
    // corrupt scheduler value must not index the table or the cache out of bounds)
    const int Sk = PAGED ? min(max(p.context_lens[b], 0), p.max_blocks * p.block_size) : p.Sk;
    const int row = rt * 32 + l31;                 // packed row of this lane inside the unit
    const bool valid = row < g * Sq;
    const int head = hk * g + (valid ? row / Sq : 0), qi = valid ? row % Sq : 0;

    const size_t kvoff = PAGED ? 0 : (size_t)(b * p.Hkv + hk) * Sk * RB;
    const __amdgpu_buffer_rsrc_t krs = skv_srd(reinterpret_cast<const char*>(p.k) + kvoff, PAGED ? 0u : (unsigned)Sk * RB);
    const __amdgpu_buffer_rsrc_t vrs = skv_srd(reinterpret_cast<const char*>(p.v) + kvoff, PAGED ? 0u : (unsigned)Sk * RB);
    // paged: byte address of key/value row `kv` of this unit inside the cache (64-bit: caches exceed 4 GiB)
    const int* const bt = PAGED ? p.block_tables + (size_t)b * p.max_blocks : nullptr;
    auto paged_row = [&](int kv) -> size_t {
        const int lb = kv / p.block_size, off = kv - lb * p.block_size;
        const size_t phys = (size_t)bt[min(lb, p.max_blocks - 1)];   // (tiles are rounded up: rows past Sk are masked, never out of the table)
        return ((phys * p.block_size + off) * p.Hkv + hk) * (size_t)RB;
    };

    // Q fragments (B operand of S^T = K.Q^T): lane (row, hi) holds d = 16ks + 8hi .. +7; rows beyond the unit are 0
    v8 qf[KS];
    {
        const char* qrow = reinterpret_cast<const char*>(p.q) + ((size_t)(b * p.Hq + head) * Sq + qi) * RB;
        const unsigned flip = p.negq ? 0x80008000u : 0u;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            u32x4_t x = {0u, 0u, 0u, 0u};
            if (valid) x = *reinterpret_cast<const u32x4_t*>(qrow + (2 * ks + hi) * 16);
            x[0] ^= flip; x[1] ^= flip; x[2] ^= flip; x[3] ^= flip;
            qf[ks] = as_v8<T>(x);
        }
    }

    // V staging map (sub-tiled image filled linearly by lane id) and transpose-read offset: fa_fwd_pp_gfx950.hip
    int v_g[NV], v_row[NV], v_col[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int bidx = (lane >> 3) + 8 * i;  // sub-tile index = kv4 * (D/16) + d16
        v_row[i] = (bidx / (D / 16)) * 4 + ((lane >> 1) & 3);
        v_col[i] = ((bidx % (D / 16)) * 2 + (lane & 1)) * 16;
        v_g[i] = v_row[i] * RB + v_col[i];
    }
    const int tr_off = hi * (D / 16) * 128 + ((lane >> 4) & 1) * 128 + (lane & 15) * 8;
    // paged fast path (power-of-two block size >= 8): block index inside the tile and byte offset inside the block
    const bool pow2 = PAGED && p.block_size >= 8 && (p.block_size & (p.block_size - 1)) == 0;
    const int bs_log2 = PAGED ? 31 - __builtin_clz(p.block_size | 1) : 0;
    const size_t blk_bytes = PAGED ? (size_t)p.block_size * p.Hkv * RB : 0;
    int k_jb = 0, k_off = 0, v_jb[NV], v_off[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) { v_jb[i] = 0; v_off[i] = 0; }
    if (pow2) {
        k_jb = l31 >> bs_log2;
        k_off = ((l31 & (p.block_size - 1)) * p.Hkv + hk) * RB + hi * 16;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            v_jb[i] = v_row[i] >> bs_log2;
            v_off[i] = ((v_row[i] & (p.block_size - 1)) * p.Hkv + hk) * RB + v_col[i];
        }
    }

    f32x16_t o[DB];
#pragma unroll
    for (int d = 0; d < DB; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
    float m = -INFINITY, l = 0.f;
    const float c = p.c;

    const int ntiles = (Sk + 31) / 32;
    int t0 = (blockIdx.x * 4 + wave) * p.chunk_tiles;
    const int t1 = min(t0 + p.chunk_tiles, ntiles);
    if constexpr (PAGED) {
        if (p.window > 0) t0 = max(t0, max(0, Sk - p.window) / 32);   // tiles entirely before the window
    }
    f32x16_t z;
#pragma unroll
    for (int r = 0; r < 16; ++r) z[r] = 0.f;

    for (int t = t0; t < t1; ++t) {
        const int kv0 = t * 32;
        u32x4_t ka[KS], vx[NV];
        if constexpr (PAGED) {
            const u32x4_t zero = {0u, 0u, 0u, 0u};
            if (pow2) {
                // power-of-two block sizes >= 8 (the usual 16/32/64/128): the tile's <= 4 logical blocks are looked up
                // ONCE per tile with wave-uniform (scalar) loads; each lane picks its block with selects and adds a
                // 32-bit in-block offset computed once per launch -- no per-lane table lookups or 64-bit multiplies
                const int lb0 = kv0 >> bs_log2;
                size_t pb[4];
                const size_t tile_off = (size_t)(kv0 & (p.block_size - 1)) * p.Hkv * RB;   // blocks larger than a tile
#pragma unroll
                for (int jb = 0; jb < 4; ++jb) pb[jb] = (size_t)bt[min(lb0 + jb, p.max_blocks - 1)] * blk_bytes + tile_off;
                auto pick = [&](int jb) -> size_t { return jb == 0 ? pb[0] : (jb == 1 ? pb[1] : (jb == 2 ? pb[2] : pb[3])); };
                const bool kin = kv0 + l31 < Sk;
                const char* krow = reinterpret_cast<const char*>(p.k) + pick(k_jb) + k_off;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) ka[ks] = kin ? *reinterpret_cast<const u32x4_t*>(krow + ks * 32) : zero;
#pragma unroll
                for (int i = 0; i < NV; ++i) {
                    const bool vin = kv0 + v_row[i] < Sk;
                    vx[i] = vin ? *reinterpret_cast<const u32x4_t*>(reinterpret_cast<const char*>(p.v) + pick(v_jb[i]) + v_off[i]) : zero;
                }
            } else {
            const bool kin = kv0 + l31 < Sk;
            const char* krow = reinterpret_cast<const char*>(p.k) + (kin ? paged_row(kv0 + l31) : 0);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) ka[ks] = kin ? *reinterpret_cast<const u32x4_t*>(krow + (2 * ks + hi) * 16) : zero;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const bool vin = kv0 + v_row[i] < Sk;
                vx[i] = vin ? *reinterpret_cast<const u32x4_t*>(reinterpret_cast<const char*>(p.v) + paged_row(kv0 + v_row[i]) + v_col[i]) : zero;
            }
            }
        } else {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
                ka[ks] = __builtin_amdgcn_raw_buffer_load_b128(krs, (kv0 + l31) * RB + (2 * ks + hi) * 16, 0, 0);
#pragma unroll
            for (int i = 0; i < NV; ++i) vx[i] = __builtin_amdgcn_raw_buffer_load_b128(vrs, v_g[i], kv0 * RB, 0);
        }
        f32x16_t s;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) s = T::mfma(as_v8<T>(ka[ks]), qf[ks], ks == 0 ? z : s);
#pragma unroll
        for (int i = 0; i < NV; ++i) *reinterpret_cast<u32x4_t*>(Vw + lane * 16 + i * 1024) = vx[i];

        // online softmax over this tile's 32 keys (16 per lane half), exp2 domain
        const bool ragged = kv0 + 32 > Sk || (PAGED && p.window > 0 && Sk - 1 - kv0 >= p.window);
        float mx = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float x = s[r] * c;
            if (ragged) {
                const int kv = kv0 + crow(r, hi);
                if (kv >= Sk || (PAGED && p.window > 0 && Sk - 1 - kv >= p.window)) x = -INFINITY;
            }
            s[r] = x;
            mx = fmaxf(mx, x);
        }
        mx = fmaxf(mx, xhalf(mx));
        const float m_new = fmaxf(m, mx);   // finite: every tile has at least one key < Sk
        const float alpha = fast_exp2(m - m_new);
        m = m_new;
        float ls = 0.f;
        u32x4_t pu[2];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float p0 = fast_exp2(s[2 * i] - m_new), p1 = fast_exp2(s[2 * i + 1] - m_new);
            ls += p0 + p1;
            pu[i >> 2][i & 3] = T::pack2(p0, p1);
        }
        l = l * alpha + ls;
#pragma unroll
        for (int d = 0; d < DB; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[d][r] *= alpha;
        // O^T += V^T . P^T  (A by transpose read from the wave's LDS tile; k-slot order = S accumulator order)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int d = 0; d < DB; ++d) {
                const int off = ((4 * kk) * (D / 16) + 2 * d) * 128;
                const s16x4_t a0 = lds_tr16(Vw + tr_off + off);
                const s16x4_t a1 = lds_tr16(Vw + tr_off + off + 2 * (D / 16) * 128);
                o[d] = T::mfma(as_v8<T>(a0, a1), as_v8<T>(pu[kk]), o[d]);
            }
    }

    // partial of this wave: O (un-normalised), m, l of the lane's row (both lane halves hold the same row)
    const float lt = l + xhalf(l);
    const int pi = blockIdx.x * 4 + wave;
    const size_t prow = (size_t)pi * p.rows_total + (size_t)blockIdx.y * 32 + l31;
    float* dst = p.part + prow * (D + 2);
#pragma unroll
    for (int d = 0; d < DB; ++d)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            const f32x4_t x = {o[d][4 * g4], o[d][4 * g4 + 1], o[d][4 * g4 + 2], o[d][4 * g4 + 3]};
            *reinterpret_cast<f32x4_t*>(dst + 32 * d + 8 * g4 + 4 * hi) = x;
        }
    if (hi == 0) {
        dst[D] = m;
        dst[D + 1] = lt;
    }
}

// One workgroup per packed row: log-sum-exp merge of the row's partials (there can be hundreds -- a serial loop per
// thread made this kernel, not the split kernel, the bottleneck: 230 us for C5b), cast, LSE.  Threads are arranged as
// G groups x D/4 column chunks; group j merges partials j, j+G, ...; the groups are then summed through LDS.
template <class T, int D>
__global__ void __launch_bounds__(256) fa_fwd_splitkv_combine(const SplitParams p) {
    constexpr int C4 = D / 4, G = 256 / C4;
    __shared__ float red[256];
    __shared__ __attribute__((aligned(16))) float accs[G][D];
    __shared__ float lsum[G];
    const int prow = blockIdx.x, tid = threadIdx.x;
    const int g = p.Hq / p.Hkv;
    const int unit = prow / (p.nrt * 32), row = prow % (p.nrt * 32);
    if (row >= g * p.Sq) return;
    const int b = unit / p.Hkv, hk = unit % p.Hkv, head = hk * g + row / p.Sq, qi = row % p.Sq;
    const float* base = p.part + (size_t)prow * (D + 2);
    const size_t pstride = (size_t)p.rows_total * (D + 2);
    // M = max over the partials' maxima
    float mx = -INFINITY;
    for (int i = tid; i < p.npart; i += 256) mx = fmaxf(mx, base[i * pstride + D]);
    red[tid] = mx;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if (tid < st) red[tid] = fmaxf(red[tid], red[tid + st]);
        __syncthreads();
    }
    const float M = red[0];
    const int grp = tid / C4, c4 = tid % C4;
    f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
    float L = 0.f;
#pragma unroll 8
    for (int i = grp; i < p.npart; i += G) {
        const float* src = base + i * pstride;
        const float w = (M == -INFINITY) ? 0.f : fast_exp2(src[D] - M);   // empty partial: m = -inf -> 0
        const f32x4_t x = *reinterpret_cast<const f32x4_t*>(src + 4 * c4);
        acc += x * w;
        if (c4 == 0) L += w * src[D + 1];
    }
    *reinterpret_cast<f32x4_t*>(&accs[grp][4 * c4]) = acc;
    if (c4 == 0) lsum[grp] = L;
    __syncthreads();
    if (tid < C4) {
        f32x4_t t = {0.f, 0.f, 0.f, 0.f};
        float Lt = 0.f;
#pragma unroll
        for (int j = 0; j < G; ++j) {
            t += *reinterpret_cast<const f32x4_t*>(&accs[j][4 * tid]);
            Lt += lsum[j];
        }
        const float inv = Lt > 0.f ? 1.0f / Lt : 0.f;   // paged: a sequence with context_len 0 has no key -> O = 0
        const size_t orow = ((size_t)(b * p.Hq + head) * p.Sq + qi);
        u32x2_t u;
        u[0] = T::pack2(t[0] * inv, t[1] * inv);
        u[1] = T::pack2(t[2] * inv, t[3] * inv);
        *reinterpret_cast<u32x2_t*>(reinterpret_cast<char*>(p.o) + orow * (D * 2) + tid * 8) = u;
        if (tid == 0 && p.lse != nullptr) p.lse[orow] = (M + fast_log2(Lt)) * kLn2;
    }
}

template <class T, int D>
int launch_split(const FwdArgs& a, hipStream_t stream) {
    SplitParams p;
    p.q = a.q; p.k = a.k; p.v = a.v; p.o = a.o; p.lse = a.lse;
    p.B = a.B; p.Hq = a.Hq; p.Hkv = a.Hkv; p.Sq = a.Sq; p.Sk = a.Sk;
    float c = a.scale * kLog2e;
    p.negq = c < 0.f;
    p.c = c < 0.f ? -c : c;
    if (p.c == 0.f) p.c = 1e-30f;
    const int g = a.Hq / a.Hkv;
    p.nrt = (g * a.Sq + 31) / 32;
    const int units = a.B * a.Hkv * p.nrt;
    const int ntiles = (a.Sk + 31) / 32;
    const int want_waves = (2048 + units - 1) / units;            // ~8 waves per CU over the whole launch
    p.chunk_tiles = (ntiles + want_waves - 1) / want_waves;
    if (p.chunk_tiles < 1) p.chunk_tiles = 1;
    const int nwaves = (ntiles + p.chunk_tiles - 1) / p.chunk_tiles;
    const int nsplit = (nwaves + 3) / 4;
    p.npart = nsplit * 4;
    p.rows_total = units * 32;
    p.block_tables = nullptr; p.context_lens = nullptr; p.block_size = 0; p.max_blocks = 0; p.window = 0;
    const size_t bytes = (size_t)p.npart * p.rows_total * (D + 2) * sizeof(float);
    if (a.query_ws != nullptr) {
        *a.query_ws = bytes;
        return 0;
    }
    ScopedWorkspace ws(bytes, a.ws, a.ws_bytes, stream);   // caller's buffer, or stream-ordered (safe with concurrent streams)
    if (ws.err != hipSuccess) return (int)ws.err;
    p.part = static_cast<float*>(ws.ptr);
    hipLaunchKernelGGL((fa_fwd_splitkv_kernel<T, D, false>), dim3((unsigned)nsplit, (unsigned)units), dim3(256), 0, stream, p);
    hipLaunchKernelGGL((fa_fwd_splitkv_combine<T, D>), dim3((unsigned)p.rows_total), dim3(256), 0, stream, p);
    return (int)hipGetLastError();
}

// Paged decode: one query token per sequence, K/V gathered through the block table; the key range is bounded by
// max_blocks * block_size on the host (no device->host sync for max(context_lens)); waves past a sequence's
// context_len leave an empty partial.
template <class T, int D>
int launch_paged(const PagedArgs& a, hipStream_t stream) {
    SplitParams p;
    p.q = a.q; p.k = a.k_cache; p.v = a.v_cache; p.o = a.out; p.lse = nullptr;
    p.B = a.B; p.Hq = a.Hq; p.Hkv = a.Hkv; p.Sq = 1; p.Sk = a.max_blocks * a.block_size;
    float c = a.scale * kLog2e;
    p.negq = c < 0.f;
    p.c = c < 0.f ? -c : c;
    if (p.c == 0.f) p.c = 1e-30f;
    const int g = a.Hq / a.Hkv;
    p.nrt = (g + 31) / 32;
    const int units = a.B * a.Hkv * p.nrt;
    const int ntiles = (p.Sk + 31) / 32;
    const int want_waves = (2048 + units - 1) / units;
    p.chunk_tiles = (ntiles + want_waves - 1) / want_waves;
    if (p.chunk_tiles < 1) p.chunk_tiles = 1;
    const int nwaves = (ntiles + p.chunk_tiles - 1) / p.chunk_tiles;
    const int nsplit = (nwaves + 3) / 4;
    p.npart = nsplit * 4;
    p.rows_total = units * 32;
    p.block_tables = a.block_tables; p.context_lens = a.context_lens;
    p.block_size = a.block_size; p.max_blocks = a.max_blocks; p.window = a.window > 0 ? a.window : 0;
    const size_t bytes = (size_t)p.npart * p.rows_total * (D + 2) * sizeof(float);
    if (a.query_ws != nullptr) {
        *a.query_ws = bytes;
        return 0;
    }
    ScopedWorkspace ws(bytes, a.ws, a.ws_bytes, stream);
    if (ws.err != hipSuccess) return (int)ws.err;
    p.part = static_cast<float*>(ws.ptr);
    hipLaunchKernelGGL((fa_fwd_splitkv_kernel<T, D, true>), dim3((unsigned)nsplit, (unsigned)units), dim3(256), 0, stream, p);
    hipLaunchKernelGGL((fa_fwd_splitkv_combine<T, D>), dim3((unsigned)p.rows_total), dim3(256), 0, stream, p);
    return (int)hipGetLastError();
}

}  // namespace

int launch_paged_decode(const PagedArgs& a, hipStream_t stream) {
    if (a.dtype == kBF16) {
        if (a.D == 128) return launch_paged<Bf16Traits, 128>(a, stream);
        if (a.D == 64) return launch_paged<Bf16Traits, 64>(a, stream);
        if (a.D == 32) return launch_paged<Bf16Traits, 32>(a, stream);
    } else if (a.dtype == kF16) {
        if (a.D == 128) return launch_paged<F16Traits, 128>(a, stream);
        if (a.D == 64) return launch_paged<F16Traits, 64>(a, stream);
        if (a.D == 32) return launch_paged<F16Traits, 32>(a, stream);
    }
    return -1;
}

#ifndef AULE_SPLITKV_MAX_UNITS
#define AULE_SPLITKV_MAX_UNITS 128   // (A/B builds override it: tools/split_ab.py)
#endif
// Shapes the split-KV path takes over from the tiled kernels: 16-bit, non-causal, no window, short queries against
// long K/V -- few enough Q blocks that the tiled kernels would leave most CUs idle.
bool splitkv_applicable(const FwdArgs& a) {
    if (a.dtype != kBF16 && a.dtype != kF16) return false;
    if (a.causal || a.window > 0) return false;
    if (a.D != 32 && a.D != 64 && a.D != 128) return false;
    if (a.Sq > 64 || a.Sk < 1024) return false;
    const long long tiled_wgs = (long long)a.B * a.Hq * ((a.Sq + 255) / 256);
    if (tiled_wgs >= 512) return false;   // two workgroups per CU: the tiled kernel fills the chip
    // K/V is streamed once per 32-row tile of packed rows, so `units` of them re-read it that many times and every
    // one adds partials to combine.  Measured (tools/split_grid.py, bf16 D128, Sk 2048 / 8192): units <= 128 wins
    // in every case (2-8x at B = 1); units = 256 is break-even (-20..25 % at Sk 2048, +4..7 % at 8192); units >= 512
    // loses (B8 Hq32 Hkv8 Sq64 Sk8192: 318 vs 194 us).
    const int g = a.Hq / a.Hkv;
    const long long units = (long long)a.B * a.Hkv * ((g * a.Sq + 31) / 32);
    return units <= AULE_SPLITKV_MAX_UNITS;
}

// Combine for FEW partials per row (the tiled kernel's SPLIT instances leave at most a few dozen): one WAVE per packed
// row, four rows per workgroup, no LDS and no block-wide reduction.  Lane i first weighs partial i (w_i = 2^(m_i - M),
// M the row's maximum; wave reductions), then every lane accumulates its D/64 columns over the partials with w_i
// broadcast from lane i.  The one-workgroup-per-row kernel above spends 256 threads and two block reductions on a row
// with 4 partials: 26 us for B8 Hq32 Hkv8 Sq64 (34 MB, 1.3 TB/s) -- a quarter of that shape's time.  Used for <= 16
// partials per row only: beyond that the serial walk over the partials is slower than the kernel above.
template <class T, int D>
__global__ void __launch_bounds__(256) fa_fwd_splitkv_combine_rows(const SplitParams p) {
    constexpr int CPL = (D + 63) / 64;            // columns per lane (D = 32: lanes 32..63 carry no column)
    const int lane = threadIdx.x & 63;
    const int prow = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (prow >= p.rows_total) return;
    const int g = p.Hq / p.Hkv;
    const int unit = prow / (p.nrt * 32), row = prow % (p.nrt * 32);
    if (row >= g * p.Sq) return;
    const int b = unit / p.Hkv, hk = unit % p.Hkv, head = hk * g + row / p.Sq, qi = row % p.Sq;
    const float* base = p.part + (size_t)prow * (D + 2);
    const size_t pstride = (size_t)p.rows_total * (D + 2);
    float acc[CPL];
#pragma unroll
    for (int c = 0; c < CPL; ++c) acc[c] = 0.f;
    float L = 0.f, M = -INFINITY;
    const bool has_col = lane * CPL < D;
    for (int i0 = 0; i0 < p.npart; i0 += 64) {     // (one trip unless there are more than 64 partials)
        const int n = min(64, p.npart - i0);
        const float mi = lane < n ? base[(size_t)(i0 + lane) * pstride + D] : -INFINITY;
        const float li = lane < n ? base[(size_t)(i0 + lane) * pstride + D + 1] : 0.f;
        float mx = mi;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
        const float Mn = fmaxf(M, mx);
        const float rescale = (M == -INFINITY) ? 0.f : fast_exp2(M - Mn);   // earlier trips (if any) move to the new maximum
        const float w = (mi == -INFINITY) ? 0.f : fast_exp2(mi - Mn);       // empty partial: weight 0
        float wl = w * li;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) wl += __shfl_xor(wl, o, 64);
        L = L * rescale + wl;
#pragma unroll
        for (int c = 0; c < CPL; ++c) acc[c] *= rescale;
        M = Mn;
        for (int i = 0; i < n; ++i) {
            const float wi = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, w), i));
            if (has_col) {
                const float* src = base + (size_t)(i0 + i) * pstride + lane * CPL;
#pragma unroll
                for (int c = 0; c < CPL; ++c) acc[c] += wi * src[c];
            }
        }
    }
    const float inv = L > 0.f ? 1.0f / L : 0.f;
    const size_t orow = ((size_t)(b * p.Hq + head) * p.Sq + qi);
    if (has_col) {
        if constexpr (CPL == 2) {
            reinterpret_cast<unsigned*>(p.o)[orow * (D / 2) + lane] = T::pack2(acc[0] * inv, acc[1] * inv);
        } else {
            const unsigned u = T::pack2(acc[0] * inv, 0.f);
            reinterpret_cast<unsigned short*>(p.o)[orow * D + lane] = (unsigned short)(u & 0xffffu);
        }
    }
    if (lane == 0 && p.lse != nullptr) p.lse[orow] = L > 0.f ? (M + fast_log2(L)) * kLn2 : -INFINITY;
}

// The combine pass on its own, for partials written by another kernel (fa_fwd_pp_gfx950.hip SPLIT instances) in
// the same layout: part [npart][B*Hkv*nrt*32][D + 2] fp32, packed row r of a unit = (head r / Sq of the group, query r % Sq).
template <class T, int D>
static int combine_only(const FwdArgs& a, float* part, int npart, int nrt, hipStream_t stream) {
    SplitParams p;
    p.q = a.q; p.k = a.k; p.v = a.v; p.o = a.o; p.lse = a.lse; p.part = part;
    p.B = a.B; p.Hq = a.Hq; p.Hkv = a.Hkv; p.Sq = a.Sq; p.Sk = a.Sk;
    p.c = 1.f; p.negq = 0; p.nrt = nrt; p.chunk_tiles = 0; p.npart = npart;
    p.rows_total = a.B * a.Hkv * nrt * 32;
    p.block_tables = nullptr; p.context_lens = nullptr; p.block_size = 0; p.max_blocks = 0; p.window = 0;
    static const int lean = [] {   // AULE_HIP_FWD_COMBINE=wg selects the workgroup-per-row kernel (A/B measurements)
        const char* e = getenv("AULE_HIP_FWD_COMBINE");
        return (e != nullptr && e[0] == 'w') ? 0 : 1;
    }();
    // Same-box A/B (tools/combine_ab.py, two passes): with <= 16 partials per row the wave-per-row kernel is ahead or
    // level (B8 Hq32 Hkv8 Sq64 Sk8192 101 -> 87 us, B4 Sq128 Sk4096 65.5 -> 52.4 us); with >= 32 its serial walk over the
    // partials loses to the kernel that spreads them over thread groups (C5b 21.7 -> 29.0 us, C5c 29.8 -> 36.2 us).
    if (lean && npart <= 16)
        hipLaunchKernelGGL((fa_fwd_splitkv_combine_rows<T, D>), dim3((unsigned)((p.rows_total + 3) / 4)), dim3(256), 0, stream, p);
    else
        hipLaunchKernelGGL((fa_fwd_splitkv_combine<T, D>), dim3((unsigned)p.rows_total), dim3(256), 0, stream, p);
    return (int)hipGetLastError();
}

int launch_splitkv_combine(const FwdArgs& a, float* part, int npart, int nrt, hipStream_t stream) {
    if (a.dtype == kBF16) {
        if (a.D == 128) return combine_only<Bf16Traits, 128>(a, part, npart, nrt, stream);
        if (a.D == 64) return combine_only<Bf16Traits, 64>(a, part, npart, nrt, stream);
        if (a.D == 32) return combine_only<Bf16Traits, 32>(a, part, npart, nrt, stream);
    } else if (a.dtype == kF16) {
        if (a.D == 128) return combine_only<F16Traits, 128>(a, part, npart, nrt, stream);
        if (a.D == 64) return combine_only<F16Traits, 64>(a, part, npart, nrt, stream);
        if (a.D == 32) return combine_only<F16Traits, 32>(a, part, npart, nrt, stream);
    }
    return -1;
}

int launch_fwd_splitkv(const FwdArgs& a, hipStream_t stream) {
    if (a.dtype == kBF16) {
        if (a.D == 128) return launch_split<Bf16Traits, 128>(a, stream);
        if (a.D == 64) return launch_split<Bf16Traits, 64>(a, stream);
        if (a.D == 32) return launch_split<Bf16Traits, 32>(a, stream);
    } else if (a.dtype == kF16) {
        if (a.D == 128) return launch_split<F16Traits, 128>(a, stream);
        if (a.D == 64) return launch_split<F16Traits, 64>(a, stream);
        if (a.D == 32) return launch_split<F16Traits, 32>(a, stream);
    }
    return -1;
}

}  // namespace aule_hip
