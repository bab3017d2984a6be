// tools/probe_layouts.hip -- hardware self-check of the three layout facts the kernels rely on
// (fa_device.h header): MFMA 32x32x16 bf16 operand/result maps, MFMA 32x32x2 f32 maps, and the
// ds_read_b64_tr_b16 gather.  Build: hipcc --offload-arch=gfx950 -O2 tools/probe_layouts.hip -o build/probe_layouts
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__global__ void k_mfma16(const float* A, const float* B, float* D) {  // A[32][16], B[16][32] row-major fp32
    int l = threadIdx.x, i = l & 31, hi = l >> 5;
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)A[i * 16 + 8 * hi + j]; b[j] = (__bf16)B[(8 * hi + j) * 32 + i]; }
    f32x16 c = {};
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * hi) * 32 + i] = c[r];
}
__global__ void k_mfma2(const float* A, const float* B, float* D) {  // A[32][2], B[2][32]
    int l = threadIdx.x, i = l & 31, hi = l >> 5;
    f32x16 c = {};
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(A[i * 2 + hi], B[hi * 32 + i], c, 0, 0, 0);
    for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * hi) * 32 + i] = c[r];
}
__global__ void k_tr(short* out) {
    __shared__ __attribute__((aligned(16))) short lds[256];
    int l = threadIdx.x;
    for (int j = 0; j < 4; ++j) lds[l * 4 + j] = (short)(l * 4 + j);
    __syncthreads();
    s16x4 t = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(lds + l * 4));
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = t[j];
}
int main() {
    int bad = 0;
    {
        std::vector<float> A(32 * 16), B(16 * 32), D(32 * 32), R(32 * 32, 0.f);
        for (int i = 0; i < 32; ++i) for (int k = 0; k < 16; ++k) A[i * 16 + k] = (float)((i * 7 + k * 3) % 11 - 5);
        for (int k = 0; k < 16; ++k) for (int n = 0; n < 32; ++n) B[k * 32 + n] = (float)((k * 5 + n * 2 + (k * n) % 3) % 13 - 6);
        for (int i = 0; i < 32; ++i) for (int n = 0; n < 32; ++n) for (int k = 0; k < 16; ++k) R[i * 32 + n] += A[i * 16 + k] * B[k * 32 + n];
        float *dA, *dB, *dD;
        hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dD, D.size() * 4);
        hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
        k_mfma16<<<1, 64>>>(dA, dB, dD);
        hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
        int e = 0; for (int x = 0; x < 1024; ++x) e += D[x] != R[x];
        printf("mfma_f32_32x32x16_bf16 layout: %s (%d mismatches)\n", e ? "FAIL" : "OK", e); bad += e != 0;
        std::vector<float> A2(64), B2(64), R2(1024, 0.f);
        for (int i = 0; i < 32; ++i) for (int k = 0; k < 2; ++k) { A2[i * 2 + k] = (float)(i * 3 + k * 17 - 20); B2[k * 32 + i] = (float)((i * i) % 7 + k * 5 - 3); }
        for (int i = 0; i < 32; ++i) for (int n = 0; n < 32; ++n) for (int k = 0; k < 2; ++k) R2[i * 32 + n] += A2[i * 2 + k] * B2[k * 32 + n];
        hipMemcpy(dA, A2.data(), 64 * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B2.data(), 64 * 4, hipMemcpyHostToDevice);
        k_mfma2<<<1, 64>>>(dA, dB, dD);
        hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
        e = 0; for (int x = 0; x < 1024; ++x) e += D[x] != R2[x];
        printf("mfma_f32_32x32x2_f32 layout: %s (%d mismatches)\n", e ? "FAIL" : "OK", e); bad += e != 0;
    }
    {
        short* d; hipMalloc(&d, 512); std::vector<short> o(256);
        k_tr<<<1, 64>>>(d);
        hipMemcpy(o.data(), d, 512, hipMemcpyDeviceToHost);
        int e = 0;
        for (int l = 0; l < 64; ++l) for (int j = 0; j < 4; ++j) {
            int want = (l & 15) + j * 16 + (l >> 4) * 64;   // fa_device.h: element (c&3) of lane 4j+(c>>2)
            e += o[l * 4 + j] != want;
        }
        printf("ds_read_b64_tr_b16 gather: %s (%d mismatches)\n", e ? "FAIL" : "OK", e); bad += e != 0;
        if (e) for (int l = 0; l < 64; ++l) printf("lane %2d: %4d %4d %4d %4d\n", l, o[l*4], o[l*4+1], o[l*4+2], o[l*4+3]);
    }
    hipError_t err = hipDeviceSynchronize();
    printf("probe done: %s, hip=%s\n", bad ? "FAIL" : "ALL OK", hipGetErrorString(err));
    return bad;
}
