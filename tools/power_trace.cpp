// tools/power_trace.cpp -- socket power, power limit and shader clock WHILE the attention kernels run.
//   hipcc -O2 --offload-arch=gfx950 -Iinclude tools/power_trace.cpp -o build/power_trace -ldl -lpthread
//   build/power_trace <lib.so> <fwd|bwd> B Hq Hkv Sq Sk D <bf16|fp16> <causal> <seconds> <amp> [label]
// Launches the call back to back through the C-ABI for `seconds` of wall time with N(0, amp) inputs (amp 0: all-zero inputs) while
// tools/power_sampler.h reads the hwmon / pp_dpm files every 10 ms; prints the mean launch time per 100-launch window next to the
// telemetry.  One leg per process (the kernel-choice switches are read once per process: AULE_HIP_FWD_KERNEL=pp, AULE_HIP_BWD_MODE=spill etc.).
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "aule.h"
#include "power_sampler.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

__global__ void fill16(uint16_t* p, size_t n, uint32_t seed, int bf16, float amp) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint32_t h = (uint32_t)i * 2654435761u ^ seed;
        float acc = 0.f;
        for (int k = 0; k < 4; ++k) { h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16; acc += (float)(h >> 8) * (1.0f / 16777216.0f) - 0.5f; }
        const float v = acc * 1.7320508f * amp;
        if (bf16) { uint32_t u = __float_as_uint(v); u += 0x7fffu + ((u >> 16) & 1u); p[i] = (uint16_t)(u >> 16); }
        else { const _Float16 hf = (_Float16)v; p[i] = *reinterpret_cast<const uint16_t*>(&hf); }
    }
}

int main(int argc, char** argv) {
    if (argc < 13) { fprintf(stderr, "usage: power_trace lib fwd|bwd B Hq Hkv Sq Sk D bf16|fp16 causal seconds amp [label]\n"); return 1; }
    void* lib = dlopen(argv[1], RTLD_NOW | RTLD_LOCAL);
    if (!lib) { fprintf(stderr, "dlopen: %s\n", dlerror()); return 1; }
    auto init = (int32_t (*)())dlsym(lib, "aule_init");
    auto fwd = (int32_t (*)(const aule_attn_desc*))dlsym(lib, "aule_attention_forward_ex");
    auto bwd = (int32_t (*)(const aule_attn_bwd_desc*))dlsym(lib, "aule_attention_backward_ex");
    auto wsz = (uint64_t (*)(const aule_attn_bwd_desc*))dlsym(lib, "aule_attention_backward_workspace_size");
    auto err = (const char* (*)())dlsym(lib, "aule_get_error");
    if (!init || !fwd || !bwd || !wsz) { fprintf(stderr, "symbols missing\n"); return 1; }
    const bool do_bwd = argv[2][0] == 'b';
    const uint32_t B = atoi(argv[3]), Hq = atoi(argv[4]), Hkv = atoi(argv[5]), Sq = atoi(argv[6]), Sk = atoi(argv[7]), D = atoi(argv[8]);
    const int bf16 = argv[9][0] == 'b';
    const int causal = atoi(argv[10]);
    const double seconds = atof(argv[11]);
    const float amp = (float)atof(argv[12]);
    const char* label = argc > 13 ? argv[13] : "leg";
    if (init() != 0) { fprintf(stderr, "aule_init: %s\n", err ? err() : "?"); return 1; }
    const size_t nq = (size_t)B * Hq * Sq * D, nk = (size_t)B * Hkv * Sk * D, nl = (size_t)B * Hq * Sq;
    uint16_t *q, *k, *v, *o, *dout, *dq, *dk, *dv; float* lse; void* ws = nullptr;
    CK(hipMalloc(&q, nq * 2)); CK(hipMalloc(&k, nk * 2)); CK(hipMalloc(&v, nk * 2)); CK(hipMalloc(&o, nq * 2)); CK(hipMalloc(&dout, nq * 2));
    CK(hipMalloc(&dq, nq * 2)); CK(hipMalloc(&dk, nk * 2)); CK(hipMalloc(&dv, nk * 2)); CK(hipMalloc(&lse, nl * 4));
    fill16<<<1024, 256>>>(q, nq, 0x1234u, bf16, amp); fill16<<<1024, 256>>>(k, nk, 0x5678u, bf16, amp);
    fill16<<<1024, 256>>>(v, nk, 0x9abcu, bf16, amp); fill16<<<1024, 256>>>(dout, nq, 0xdef0u, bf16, amp);
    CK(hipDeviceSynchronize());
    aule_attn_desc fd; memset(&fd, 0, sizeof fd);
    fd.struct_size = sizeof fd; fd.dtype = bf16 ? AULE_DTYPE_BF16 : AULE_DTYPE_F16;
    fd.batch = B; fd.heads_q = Hq; fd.heads_kv = Hkv; fd.seq_q = Sq; fd.seq_k = Sk; fd.head_dim = D;
    fd.scale = 0.f; fd.causal = causal; fd.window_size = 0; fd.device = -1; fd.stream = nullptr;
    fd.q = q; fd.k = k; fd.v = v; fd.out = o; fd.lse = lse;
    aule_attn_bwd_desc bd; memset(&bd, 0, sizeof bd);
    bd.struct_size = sizeof bd; bd.dtype = fd.dtype; bd.batch = B; bd.heads_q = Hq; bd.heads_kv = Hkv; bd.seq_q = Sq; bd.seq_k = Sk; bd.head_dim = D;
    bd.scale = 0.f; bd.causal = causal; bd.window_size = 0; bd.device = -1; bd.stream = nullptr;
    bd.q = q; bd.k = k; bd.v = v; bd.out = o; bd.dout = dout; bd.lse = lse; bd.dq = dq; bd.dk = dk; bd.dv = dv;
    const uint64_t wbytes = wsz(&bd);
    if (wbytes) CK(hipMalloc(&ws, wbytes));
    bd.workspace = ws; bd.workspace_bytes = wbytes;
    int32_t rc = fwd(&fd);
    if (rc != 0) { fprintf(stderr, "forward rc %d: %s\n", rc, err ? err() : "?"); return 1; }
    CK(hipDeviceSynchronize());

    PowerSampler ps;
    ps.discover(getenv("PT_VERBOSE") != nullptr);
    // idle baseline: 300 ms of samples with nothing running
    ps.start();
    std::this_thread::sleep_for(std::chrono::milliseconds(300));
    ps.stop((std::string(label) + "/idle-before").c_str(), 0, 0);

    const int W = 100;   // launches per timing window
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<std::pair<double, double>> win;   // (t_ms at window end, us per launch)
    const auto t0 = std::chrono::steady_clock::now();
    ps.start();
    for (;;) {
        CK(hipEventRecord(e0, nullptr));
        for (int j = 0; j < W; ++j) rc = do_bwd ? bwd(&bd) : fwd(&fd);
        CK(hipEventRecord(e1, nullptr));
        if (rc != 0) { fprintf(stderr, "rc %d: %s\n", rc, err ? err() : "?"); return 1; }
        CK(hipEventSynchronize(e1));
        float t; CK(hipEventElapsedTime(&t, e0, e1));
        const double now = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        win.push_back({now, t * 1e3 / W});
        if (now >= seconds * 1e3) break;
    }
    const double flops = 4.0 * B * Hq * (double)Sq * Sk * D * (causal ? 0.5 : 1.0) * (do_bwd ? 2.5 : 1.0);
    printf("== %s: %s B%u %u/%u S%u/%u D%u %s causal%d amp %.2f, %zu windows of %d launches over %.1f s\n", label, do_bwd ? "bwd" : "fwd", B, Hq, Hkv, Sq, Sk, D,
           bf16 ? "bf16" : "fp16", causal, amp, win.size(), W, win.back().first / 1e3);
    ps.stop(label, 500.0, 10);
    printf("  launch time per window (t_ms, us per launch, TFLOP/s):\n");
    double s = 0; int n = 0;
    for (size_t i = 0; i < win.size(); ++i) {
        if (i < 8 || i % 8 == 0) printf("   %8.1f %8.1f %8.1f\n", win[i].first, win[i].second, flops / (win[i].second * 1e-6) * 1e-12);
        if (win[i].first >= 500.0) { s += win[i].second; ++n; }
    }
    if (n) printf("  mean after 500 ms: %.1f us per launch = %.1f TFLOP/s\n", s / n, flops / (s / n * 1e-6) * 1e-12);
    return 0;
}
