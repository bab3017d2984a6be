// rope_gfx950.hip -- rotary position embedding of Q / K (and the transposed rotation of dQ / dK) for MI355X.
//
// Why a pass of its own and not a branch in the attention kernels: the rotation of K must happen once per key
// row, while the tiled forward re-reads every K tile once per Q block (16 times at S = 4096) and both backward
// kernels re-read Q and K tiles -- rotating inside them would repeat the sin/cos loads and 4 VALU ops per element
// in the MFMA-bound inner loops.  As a pass it is pure HBM streaming (read x, read the table rows, write x'):
// 16-byte loads/stores, one thread per 8 (16-bit) or 4 (fp32) rotation pairs, no LDS; the table rows
// ([S, D/2] fp32, shared by every head) stay in L2.  Bound: HBM; algorithmic bytes = 2 * rows * D * sizeof(T).
//
// Semantics (both layouts of the reference):
//   layout 0 "half"        pairs (p, p + D/2): python/aule/triton_flash.py:32-52, :112-131, :165-180, :680-703
//   layout 1 "interleaved" pairs (2p, 2p + 1): shaders/attention_f32.comp:98-111 and :132-145
//   x1' = x1 cos - x2 sin,  x2' = x1 sin + x2 cos,  table row = sequence index + pos_offset,
//   computed in fp32 and rounded once to the I/O dtype; `inverse` flips the sign of sin (the transpose of the
//   rotation: what turns dQ', dK' into dQ, dK).
#include "fa_device.h"
#include "fa_kernels.h"

namespace aule_hip {
namespace {

struct RopeParams {
    const void* in;
    void* out;
    const float* cos;
    const float* sin;
    long long nrows;   // B * H * S
    int S, D, half;
    int pitch;         // elements per row of in/out
    int tpitch;        // floats per table row (>= half)
    int pos_offset;
    float sgn;         // +1, or -1 for the inverse rotation
};

template <class E> struct Conv;
template <> struct Conv<float> {
    static __device__ __forceinline__ float ld(float x) { return x; }
    static __device__ __forceinline__ float st(float x) { return x; }
};
template <> struct Conv<__bf16> {
    static __device__ __forceinline__ float ld(__bf16 x) { return (float)x; }
    static __device__ __forceinline__ __bf16 st(float x) { return (__bf16)x; }
};
template <> struct Conv<_Float16> {
    static __device__ __forceinline__ float ld(_Float16 x) { return (float)x; }
    static __device__ __forceinline__ _Float16 st(float x) { return (_Float16)x; }
};

template <class E, int V>
struct alignas(sizeof(E) * V) Pack {
    E v[V];
};

// One thread = V rotation pairs of one row.  V > 1 needs half % V == 0 and 16-byte aligned rows (checked on the
// host); V = 1 is the fallback for any even D and any pitch.
template <class E, int V, bool INTERLEAVED>
__global__ void __launch_bounds__(256) rope_kernel(const RopeParams p) {
    const int gpr = p.half / V;  // thread groups per row
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    if (gid >= p.nrows * gpr) return;
    const long long row = gid / gpr;
    const int p0 = (int)(gid % gpr) * V;
    const int s = (int)(row % p.S) + p.pos_offset;
    const Pack<float, V> c = *reinterpret_cast<const Pack<float, V>*>(p.cos + (size_t)s * p.tpitch + p0);
    const Pack<float, V> sn = *reinterpret_cast<const Pack<float, V>*>(p.sin + (size_t)s * p.tpitch + p0);
    const E* xin = static_cast<const E*>(p.in) + (size_t)row * p.pitch;
    E* xout = static_cast<E*>(p.out) + (size_t)row * p.pitch;
    if constexpr (INTERLEAVED) {
        const Pack<E, 2 * V> x = *reinterpret_cast<const Pack<E, 2 * V>*>(xin + 2 * p0);
        Pack<E, 2 * V> y;
#pragma unroll
        for (int i = 0; i < V; ++i) {
            const float x1 = Conv<E>::ld(x.v[2 * i]), x2 = Conv<E>::ld(x.v[2 * i + 1]);
            const float si = p.sgn * sn.v[i];
            float y1, y2;
            rope_pair(x1, x2, c.v[i], si, y1, y2);
            y.v[2 * i] = Conv<E>::st(y1);
            y.v[2 * i + 1] = Conv<E>::st(y2);
        }
        *reinterpret_cast<Pack<E, 2 * V>*>(xout + 2 * p0) = y;
    } else {
        const Pack<E, V> a = *reinterpret_cast<const Pack<E, V>*>(xin + p0);
        const Pack<E, V> b = *reinterpret_cast<const Pack<E, V>*>(xin + p.half + p0);
        Pack<E, V> ya, yb;
#pragma unroll
        for (int i = 0; i < V; ++i) {
            const float x1 = Conv<E>::ld(a.v[i]), x2 = Conv<E>::ld(b.v[i]);
            const float si = p.sgn * sn.v[i];
            float y1, y2;
            rope_pair(x1, x2, c.v[i], si, y1, y2);
            ya.v[i] = Conv<E>::st(y1);
            yb.v[i] = Conv<E>::st(y2);
        }
        *reinterpret_cast<Pack<E, V>*>(xout + p0) = ya;
        *reinterpret_cast<Pack<E, V>*>(xout + p.half + p0) = yb;
    }
}

template <class E, int V, bool IL>
int run(const RopeParams& p, hipStream_t stream) {
    const long long threads = p.nrows * (p.half / V);
    const long long blocks = (threads + 255) / 256;
    if (blocks <= 0) return 0;
    if (blocks > 0x7fffffffLL) return -1;
    hipLaunchKernelGGL((rope_kernel<E, V, IL>), dim3((unsigned)blocks), dim3(256), 0, stream, p);
    return (int)hipGetLastError();
}

template <class E, int VMAX>
int dispatch(const RopeParams& p, bool interleaved, hipStream_t stream) {
    // vector path: V pairs per thread as 16-byte accesses of x (half layout: V elements per access; interleaved:
    // 2V elements), table accesses of V floats
    const auto aligned = [](const void* q, size_t a) { return (reinterpret_cast<uintptr_t>(q) % a) == 0; };
    const size_t es = sizeof(E);
    if (interleaved) {
        constexpr int V = VMAX / 2;
        const bool ok = p.half % V == 0 && (p.pitch * es) % (2 * V * es) == 0 && aligned(p.in, 2 * V * es) &&
                        aligned(p.out, 2 * V * es) && aligned(p.cos, V * 4) && aligned(p.sin, V * 4) && p.tpitch % V == 0;
        return ok ? run<E, V, true>(p, stream) : run<E, 1, true>(p, stream);
    }
    constexpr int V = VMAX;
    const bool ok = p.half % V == 0 && (p.pitch * es) % (V * es) == 0 && aligned(p.in, V * es) && aligned(p.out, V * es) &&
                    aligned(p.cos, V * 4) && aligned(p.sin, V * 4) && p.tpitch % V == 0;
    return ok ? run<E, V, false>(p, stream) : run<E, 1, false>(p, stream);
}

}  // namespace

int launch_rope(const RopeArgs& a, hipStream_t stream) {
    if (a.D <= 0 || (a.D & 1) || a.S <= 0 || a.pitch < a.D || a.pos_offset < 0) return -1;
    RopeParams p;
    p.in = a.in; p.out = a.out; p.cos = a.cos; p.sin = a.sin;
    p.nrows = a.nheads * (long long)a.S;
    p.S = a.S; p.D = a.D; p.half = a.D / 2; p.pitch = a.pitch; p.pos_offset = a.pos_offset;
    p.tpitch = a.table_pitch > 0 ? a.table_pitch : p.half;
    if (p.tpitch < p.half) return -1;
    p.sgn = a.inverse ? -1.0f : 1.0f;
    const bool il = a.layout == 1;
    if (a.dtype == kF32) return dispatch<float, 4>(p, il, stream);
    if (a.dtype == kBF16) return dispatch<__bf16, 8>(p, il, stream);
    if (a.dtype == kF16) return dispatch<_Float16, 8>(p, il, stream);
    return -1;
}

}  // namespace aule_hip
