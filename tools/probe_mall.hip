// tools/probe_mall.hip -- does the 256 MB Infinity Cache keep what a kernel WROTE for the next kernel to read?  (round 5: the 5-matmul
// backward spills 0.57 .. 2.2 GB of packed dS from the dK/dV kernel and reads it back in the dQ kernel; VERDICT r4 item 1c asks for
// plain read-modify-write throughput as well.)  For N MB: kernel W writes N MB (plain or non-temporal stores); kernel R reads them
// back in the same order or in reverse (last written first); kernel M adds 1.0f to every float in place (plain RMW, second pass timed).
//   hipcc -O3 --offload-arch=gfx950 tools/probe_mall.hip -o build/probe_mall && build/probe_mall
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int NT>
__global__ void __launch_bounds__(256) kw(u32x4* p, size_t n16) {
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += stride) {
        u32x4 v = {(unsigned)i, 1u, 2u, 3u};
        if (NT) __builtin_nontemporal_store(v, p + i); else p[i] = v;
    }
}
template <int REV>
__global__ void __launch_bounds__(256) kr(const u32x4* p, size_t n16, unsigned* out) {
    const size_t stride = (size_t)gridDim.x * 256;
    u32x4 a = {0, 0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += stride) {
        const u32x4 v = p[REV ? n16 - 1 - i : i];
        a ^= v;
    }
    if ((a[0] ^ a[1] ^ a[2] ^ a[3]) == 0x12345u) out[0] = 1;
}
__global__ void __launch_bounds__(256) km(f32x4* p, size_t n16) {
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += stride) {
        f32x4 v = p[i];
        v += 1.0f;
        p[i] = v;
    }
}
// plain RMW NEXT TO MFMA load: every wave alternates `nm` v_mfma_f32_32x32x16_bf16 with one 4 KB read-add-write of fp32
// (64 lanes x 16 bytes x 4), the working set `bytes` walked by the whole grid; reports the RMW stream's GB/s (read + write) and the MFMA TFLOP/s.
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
__global__ void __launch_bounds__(256, 1) kmm(f32x4* p, size_t n16, int nm, int iters, float* out) {
    f32x16 acc[4];
    for (int d = 0; d < 4; ++d) for (int r = 0; r < 16; ++r) acc[d][r] = 0.f;
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(0.25f + 0.01f * ((threadIdx.x * 7 + j) & 31)); b[j] = (__bf16)(-0.5f + 0.02f * ((threadIdx.x * 3 + j) & 31)); }
    const size_t wave = ((size_t)blockIdx.x * 256 + threadIdx.x) >> 6, nw = (size_t)gridDim.x * 4, lane = threadIdx.x & 63;
    size_t tile = wave;                       // 4 KB tiles: 256 x 16 bytes
    const size_t ntile = n16 / 256;
    for (int it = 0; it < iters; ++it) {
        f32x4 v[4];
        for (int k = 0; k < 4; ++k) v[k] = p[(tile % ntile) * 256 + k * 64 + lane];
        for (int m = 0; m < nm; ++m) acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[m & 3], 0, 0, 0);
        for (int k = 0; k < 4; ++k) { v[k] += acc[0][k]; p[(tile % ntile) * 256 + k * 64 + lane] = v[k]; }
        tile += nw;
    }
    float s = 0.f;
    for (int d = 0; d < 4; ++d) for (int r = 0; r < 16; ++r) s += acc[d][r];
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
    const size_t maxb = (size_t)4096 << 20;
    char* buf; unsigned* out;
    hipMalloc(&buf, maxb); hipMalloc(&out, 64); hipMemset(buf, 0, maxb);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = 256 * 8;
    auto tm = [&](auto fn) { hipEventRecord(e0); fn(); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); return ms; };
    printf("%8s %12s %12s %12s %12s %12s %12s\n", "MB", "W GB/s", "R-after-W", "Rrev-after-W", "R-after-Wnt", "Rrev-aft-Wnt", "RMW GB/s(r+w)");
    for (int mb : {16, 32, 64, 96, 128, 160, 192, 256, 384, 512, 1024, 2048}) {
        const size_t bytes = (size_t)mb << 20, n16 = bytes / 16;
        double best[6] = {0, 0, 0, 0, 0, 0};
        for (int rep = 0; rep < 4; ++rep) {
            // evict: touch the other half of the buffer
            tm([&] { hipLaunchKernelGGL(kw<0>, dim3(grid), dim3(256), 0, 0, (u32x4*)(buf + (maxb >> 1)), (size_t)(768 << 20) / 16); });
            float w = tm([&] { hipLaunchKernelGGL(kw<0>, dim3(grid), dim3(256), 0, 0, (u32x4*)buf, n16); });
            float r = tm([&] { hipLaunchKernelGGL(kr<0>, dim3(grid), dim3(256), 0, 0, (const u32x4*)buf, n16, out); });
            tm([&] { hipLaunchKernelGGL(kw<0>, dim3(grid), dim3(256), 0, 0, (u32x4*)(buf + (maxb >> 1)), (size_t)(768 << 20) / 16); });
            tm([&] { hipLaunchKernelGGL(kw<0>, dim3(grid), dim3(256), 0, 0, (u32x4*)buf, n16); });
            float rr = tm([&] { hipLaunchKernelGGL(kr<1>, dim3(grid), dim3(256), 0, 0, (const u32x4*)buf, n16, out); });
            tm([&] { hipLaunchKernelGGL(kw<0>, dim3(grid), dim3(256), 0, 0, (u32x4*)(buf + (maxb >> 1)), (size_t)(768 << 20) / 16); });
            tm([&] { hipLaunchKernelGGL(kw<1>, dim3(grid), dim3(256), 0, 0, (u32x4*)buf, n16); });
            float rn = tm([&] { hipLaunchKernelGGL(kr<0>, dim3(grid), dim3(256), 0, 0, (const u32x4*)buf, n16, out); });
            tm([&] { hipLaunchKernelGGL(kw<0>, dim3(grid), dim3(256), 0, 0, (u32x4*)(buf + (maxb >> 1)), (size_t)(768 << 20) / 16); });
            tm([&] { hipLaunchKernelGGL(kw<1>, dim3(grid), dim3(256), 0, 0, (u32x4*)buf, n16); });
            float rrn = tm([&] { hipLaunchKernelGGL(kr<1>, dim3(grid), dim3(256), 0, 0, (const u32x4*)buf, n16, out); });
            tm([&] { hipLaunchKernelGGL(km, dim3(grid), dim3(256), 0, 0, (f32x4*)buf, n16); });
            float m = tm([&] { hipLaunchKernelGGL(km, dim3(grid), dim3(256), 0, 0, (f32x4*)buf, n16); });
            const float t[6] = {w, r, rr, rn, rrn, m};
            for (int k = 0; k < 6; ++k) { const double g = (k == 5 ? 2.0 : 1.0) * bytes / (t[k] * 1e-3) / 1e9; if (g > best[k]) best[k] = g; }
        }
        printf("%8d %12.0f %12.0f %12.0f %12.0f %12.0f %12.0f\n", mb, best[0], best[1], best[2], best[3], best[4], best[5]);
    }
    printf("\n# plain RMW (4 KB fp32 per wave and step) next to nm MFMAs per step, 256 workgroups x 4 waves, working set 134 MB (C3's fp32 dQ) and 1 GB:\n");
    printf("%8s %6s %14s %14s\n", "MB", "nm", "RMW GB/s(r+w)", "MFMA TFLOP/s");
    float* outm; hipMalloc(&outm, 256 * 256 * 4);
    for (int mb : {134, 1024})
        for (int nm : {0, 8, 16, 32, 64}) {
            const size_t bytes = (size_t)mb << 20, n16 = bytes / 16;
            const int iters = 4000;
            double best = 1e30;
            for (int rep = 0; rep < 3; ++rep) {
                float ms = tm([&] { hipLaunchKernelGGL(kmm, dim3(256), dim3(256), 0, 0, (f32x4*)buf, n16, nm, iters, outm); });
                if (ms < best) best = ms;
            }
            const double rmw = 2.0 * 4096.0 * 1024 * iters / (best * 1e-3) / 1e9, tf = 32768.0 * nm * 1024 * iters / (best * 1e-3) / 1e12;
            printf("%8d %6d %14.0f %14.1f\n", mb, nm, rmw, tf);
        }
    return 0;
}
