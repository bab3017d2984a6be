// probe_atomic.hip -- feasibility probe for a fused backward (dK/dV kernel that also accumulates dQ with fp32
// atomics, 5 GEMMs instead of the 7 of the two-kernel design): how fast can the chip absorb the dQ traffic of the
// C3 shape (128 heads, Sq = 2048, D = 128, 128-key blocks, causal), i.e. 272 atomic 64x128 fp32 tiles per head?
//   mode 0: no-return global_atomic_add_f32, heads XCD-local (all 16 key blocks of a head on one XCD: the dQ
//           lines of a head stay in one L2)
//   mode 1: the same atomics, key blocks of a head round-robin over the XCDs (lines bounce between L2s)
//   mode 2: plain stores of the same tiles (upper bound: write bandwidth)
//   mode 3: packed bf16 atomics (global_atomic_pk_add_bf16), XCD-local: half the bytes
// Build: hipcc -O3 --offload-arch=gfx950 tools/probe_atomic.hip -o build/probe_atomic
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <cstdio>
#include <cstdlib>

#ifndef PROBE_KB
#define PROBE_KB 128   // keys per block (round 4: -DPROBE_KB=256 = the 256-key blocks a fused kernel would need to halve the traffic)
#endif
constexpr int NH = 128, SQ = 2048, D = 128, KB = PROBE_KB, QT = 64, NKB = SQ / KB, NQT = SQ / QT;

template <int MODE>
__global__ void __launch_bounds__(256) probe(float* dq, int spin) {
    const int bid = blockIdx.x;
    int head, kb;
    if (MODE == 1) {
        kb = bid % NKB; head = bid / NKB;
    } else {  // XCD-local: xcd = bid % 8 owns heads {xcd, xcd + 8, ...}
        const int xcd = bid % 8, slot = bid / 8;      // slot: 0 .. NH/8 * NKB - 1
        head = (slot / NKB) * 8 + xcd; kb = slot % NKB;
    }
    kb = NKB - 1 - kb;   // (order does not matter here)
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    float* base = dq + (size_t)head * SQ * D;
    float acc = 0.001f * (float)(tid + 1);
    for (int qt = kb * (KB / QT); qt < NQT; ++qt) {
        // stand-in for the tile's MFMA work
        for (int s = 0; s < spin; ++s) acc = __builtin_fmaf(acc, 1.0001f, 0.5f);
        // wave w owns rows 16w .. 16w+15 of the 64-row tile; a lane adds 2 x 16 rows x (lane, lane + 64)
        float* t = base + (size_t)(qt * QT + wave * 16) * D;
        if (MODE == 3) {
            __hip_bfloat162* tb = reinterpret_cast<__hip_bfloat162*>(t);  // [row][64 pairs] (half the buffer is used)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                __hip_bfloat162 v = __float22bfloat162_rn(float2{acc, acc});
                unsafeAtomicAdd(tb + r * (D / 2) + lane, v);
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    float* a = t + r * D + h * 64 + lane;
                    if (MODE == 2) __builtin_nontemporal_store(acc, a);
                    else unsafeAtomicAdd(a, acc);
                }
        }
    }
}

int main(int argc, char** argv) {
    const int spin = argc > 1 ? atoi(argv[1]) : 0;
    float* dq;
    const size_t bytes = (size_t)NH * SQ * D * sizeof(float);
    hipMalloc(&dq, bytes);
    hipMemset(dq, 0, bytes);
    long tiles = 0;
    for (int kb = 0; kb < NKB; ++kb) tiles += NQT - kb * (KB / QT);
    const double gb = (double)NH * tiles * QT * D * 4 / 1e9;
    printf("atomic dQ traffic: %.3f GB (%ld tiles/head), spin=%d\n", gb, tiles, spin);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](auto kern, const char* name, double g) {
        for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(NH * NKB), dim3(256), 0, 0, dq, spin);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(kern, dim3(NH * NKB), dim3(256), 0, 0, dq, spin);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("  %-34s %8.1f us   %.2f TB/s\n", name, ms * 100, g / (ms / 10 * 1e-3) / 1e3);
    };
    run(probe<0>, "f32 atomics, XCD-local heads", gb);
    run(probe<1>, "f32 atomics, heads over all XCDs", gb);
    run(probe<2>, "plain stores", gb);
    run(probe<3>, "pk_bf16 atomics, XCD-local", gb / 2);
    return 0;
}
