// tools/cbench.cpp -- times aule_attention_forward_ex / aule_attention_backward_ex through the C-ABI without Python:
// a GPU box pays 1-2 minutes for its first `import torch`, this binary starts in under a second (kernel A/B loops).
//   hipcc -O2 --offload-arch=gfx950 -Iinclude tools/cbench.cpp -o build/cbench -ldl
//   build/cbench <lib.so> <fwd|bwd|tl> B Hq Hkv Sq Sk D <bf16|fp16> <causal 0|1|2> [reps=30] [warm=10] [batch=1]
// Prints the median / min of the per-launch HIP-event times and a checksum of every output (the same inputs on every run: two
// builds that should agree bit for bit print the same sums).
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "aule.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

__global__ void fill16(uint16_t* p, size_t n, uint32_t seed, int bf16, float amp) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        // two hashed uniforms -> one normal-ish value (sum of 4 uniforms, variance-matched)
        uint32_t h = (uint32_t)i * 2654435761u ^ seed;
        float acc = 0.f;
        for (int k = 0; k < 4; ++k) { h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16; acc += (float)(h >> 8) * (1.0f / 16777216.0f) - 0.5f; }
        const float v = acc * 1.7320508f * amp;
        if (bf16) { uint32_t u = __float_as_uint(v); u += 0x7fffu + ((u >> 16) & 1u); p[i] = (uint16_t)(u >> 16); }
        else { const _Float16 hf = (_Float16)v; p[i] = *reinterpret_cast<const uint16_t*>(&hf); }
    }
}

static double checksum16(const void* dev, size_t n, int bf16) {
    std::vector<uint16_t> h(n);
    if (hipMemcpy(h.data(), dev, n * 2, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    double s = 0, a = 0;
    for (size_t i = 0; i < n; ++i) {
        float v;
        if (bf16) { uint32_t u = (uint32_t)h[i] << 16; memcpy(&v, &u, 4); }
        else { _Float16 hf; memcpy(&hf, &h[i], 2); v = (float)hf; }
        s += v; a += v < 0 ? -v : v;
    }
    printf(" sum %.9g abs %.9g", s, a);
    return s;
}

int main(int argc, char** argv) {
    if (argc < 11) { fprintf(stderr, "usage: cbench lib fwd|bwd B Hq Hkv Sq Sk D bf16|fp16 causal [reps] [warm]\n"); return 1; }
    void* lib = dlopen(argv[1], RTLD_NOW | RTLD_LOCAL);
    if (!lib) { fprintf(stderr, "dlopen: %s\n", dlerror()); return 1; }
    auto init = (int32_t (*)())dlsym(lib, "aule_init");
    auto fwd = (int32_t (*)(const aule_attn_desc*))dlsym(lib, "aule_attention_forward_ex");
    auto bwd = (int32_t (*)(const aule_attn_bwd_desc*))dlsym(lib, "aule_attention_backward_ex");
    auto wsz = (uint64_t (*)(const aule_attn_bwd_desc*))dlsym(lib, "aule_attention_backward_workspace_size");
    auto err = (const char* (*)())dlsym(lib, "aule_get_error");
    if (!init || !fwd || !bwd || !wsz) { fprintf(stderr, "symbols missing\n"); return 1; }
    const bool do_bwd = argv[2][0] == 'b' || argv[2][0] == 't';
    const uint32_t B = atoi(argv[3]), Hq = atoi(argv[4]), Hkv = atoi(argv[5]), Sq = atoi(argv[6]), Sk = atoi(argv[7]), D = atoi(argv[8]);
    const int bf16 = argv[9][0] == 'b';
    const int causal = atoi(argv[10]);
    const int reps = argc > 11 ? atoi(argv[11]) : 30, warm = argc > 12 ? atoi(argv[12]) : 10;
    const int batch = argc > 13 ? atoi(argv[13]) : 1;   // calls per timed region, back to back (no host sync in between): time / batch
    if (init() != 0) { fprintf(stderr, "aule_init: %s\n", err ? err() : "?"); return 1; }
    const size_t nq = (size_t)B * Hq * Sq * D, nk = (size_t)B * Hkv * Sk * D, nl = (size_t)B * Hq * Sq;
    uint16_t *q, *k, *v, *o, *dout, *dq, *dk, *dv; float* lse; void* ws = nullptr;
    CK(hipMalloc(&q, nq * 2)); CK(hipMalloc(&k, nk * 2)); CK(hipMalloc(&v, nk * 2)); CK(hipMalloc(&o, nq * 2)); CK(hipMalloc(&dout, nq * 2));
    CK(hipMalloc(&dq, nq * 2)); CK(hipMalloc(&dk, nk * 2)); CK(hipMalloc(&dv, nk * 2)); CK(hipMalloc(&lse, nl * 4));
    const float amp = getenv("CB_AMP") ? (float)atof(getenv("CB_AMP")) : 1.f;   // CB_AMP=0: all-zero inputs (how much of the time is data-dependent power?)
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
    if (argv[2][0] == 't') {   // tl: the dK/dV kernel's timeline build (debug library, AULE_TL=dkv4): cycles per stream iteration and phase
        auto tl = (int32_t (*)(const aule_attn_bwd_desc*, unsigned long long*))dlsym(lib, "aule_hip_debug_backward_timeline");
        if (!tl) { fprintf(stderr, "not a debug library\n"); return 1; }
        unsigned long long* st; CK(hipMalloc(&st, 8 * 384 * 8)); CK(hipMemset(st, 0, 8 * 384 * 8));
        for (int i = 0; i < 10; ++i) rc = tl(&bd, st);
        CK(hipDeviceSynchronize());
        std::vector<unsigned long long> h(8 * 384); CK(hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost));
        printf("timeline rc %d\n", rc);
        for (int w = 0; w < 4; ++w) {
            const double n = (double)h[4 * w], a = (double)h[4 * w + 1], b = (double)h[4 * w + 2];
            if (n > 0) printf("  wave %d: %.0f iterations, phase 1 + boundary %7.0f  phase 2 %7.0f  = %7.0f cycles per iteration (32 MFMAs = 1024)\n", w, n, a / n, b / n, (a + b) / n);
        }
        return 0;
    }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<float> ms;
    for (int i = 0; i < warm + reps; ++i) {
        CK(hipEventRecord(e0, nullptr));
        for (int j = 0; j < batch; ++j) rc = do_bwd ? bwd(&bd) : fwd(&fd);
        CK(hipEventRecord(e1, nullptr));
        if (rc != 0) { fprintf(stderr, "rc %d: %s\n", rc, err ? err() : "?"); return 1; }
        CK(hipEventSynchronize(e1));
        float t; CK(hipEventElapsedTime(&t, e0, e1));
        if (i >= warm) ms.push_back(t / batch);
    }
    std::sort(ms.begin(), ms.end());
    const double flops = 4.0 * B * Hq * (double)Sq * Sk * D * (causal ? 0.5 : 1.0) * (do_bwd ? 2.5 : 1.0);
    printf("%s B%u %u/%u S%u/%u D%u %s causal%d: median %.1f us  min %.1f us  %.1f TF", do_bwd ? "bwd" : "fwd", B, Hq, Hkv, Sq, Sk, D, bf16 ? "bf16" : "fp16",
           causal, ms[ms.size() / 2] * 1e3, ms[0] * 1e3, flops / (ms[ms.size() / 2] * 1e-3) * 1e-12);
    printf("\n  o:"); checksum16(o, nq, bf16);
    if (do_bwd) { printf("\n  dq:"); checksum16(dq, nq, bf16); printf("\n  dk:"); checksum16(dk, nk, bf16); printf("\n  dv:"); checksum16(dv, nk, bf16); }
    printf("\n");
    return 0;
}
