// aule_capi.cpp -- extern "C" implementation of include/aule.h on the HIP runtime.
//
// Re-exports the C-ABI of the reference's src/lib.zig (global context, 1024-slot
// tensor table, 512-byte error buffer, negative return codes) over the gfx950
// kernels in this directory.  The reference's host language for this layer is
// Zig; no Zig toolchain exists in the build image, so the layer is C++ with
// identical symbol names and signatures (see INTEGRATION.md for the Zig `extern`
// block a maintainer would add to bind it).
//
// There is deliberately NO CPU fallback here: if no HIP device is present
// aule_init() fails with -1 and every compute entry point returns -1.
#include "../../include/aule.h"

#include <hip/hip_runtime_api.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstddef>
#include <cstring>
#include <mutex>

#include <dlfcn.h>

#include "fa_kernels.h"

// abi2: optional workspace / workspace_bytes appended to the forward and paged descriptors (96 -> 112, 104 -> 120)
static_assert(sizeof(aule_attn_desc) == 112 && offsetof(aule_attn_desc, lse) == 88 && offsetof(aule_attn_desc, workspace) == 96,
              "aule_attn_desc layout is part of the ABI");
static_assert(sizeof(aule_paged_desc) == 120 && offsetof(aule_paged_desc, workspace) == 104,
              "aule_paged_desc layout is part of the ABI");
static_assert(sizeof(aule_attn_bwd_desc) == 144, "aule_attn_bwd_desc layout is part of the ABI");

namespace {

using aule_hip::BwdArgs;
using aule_hip::FwdArgs;

struct DevTensor {
    void* ptr = nullptr;      // device memory, rows padded to `pitch` floats
    uint32_t shape[4] = {0, 0, 0, 0};
    uint64_t count = 0;       // logical element count (B*H*S*D)
    uint32_t pitch = 0;       // padded head_dim (32 / 64 / 128, or D itself if D > 128)
    bool used = false;
};

std::mutex g_mu;
bool g_init = false;
int g_device = 0;
uint64_t g_configured_mask = 0;  // devices on which kernel attributes were set
char g_err[512];
size_t g_err_len = 0;
DevTensor g_tensors[AULE_MAX_TENSORS];
int g_variant = 0;

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    int n = vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    if (n < 0) n = 0;
    if ((size_t)n >= sizeof(g_err)) n = sizeof(g_err) - 1;
    g_err_len = (size_t)n;
}

uint32_t pad_dim(uint32_t d) {
    if (d <= 32) return 32;
    if (d <= 64) return 64;
    if (d <= 128) return 128;
    return d;
}

struct DeviceGuard {
    int prev = -1;
    bool switched = false;
    explicit DeviceGuard(int dev) {
        if (dev < 0) return;
        if (hipGetDevice(&prev) == hipSuccess && prev != dev) {
            switched = hipSetDevice(dev) == hipSuccess;
        }
    }
    ~DeviceGuard() {
        if (switched) (void)hipSetDevice(prev);
    }
};

int ensure_configured() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return -1;
    if (dev < 64 && (g_configured_mask >> dev) & 1) return 0;
    int rc = aule_hip::configure_kernels();
    if (rc != 0) {
        set_error("Kernel configuration failed on device %d: %s", dev, hipGetErrorString((hipError_t)rc));
        return -4;
    }
    if (dev < 64) g_configured_mask |= (1ull << dev);
    return 0;
}

DevTensor* lookup(uint64_t h) {
    if (h == 0 || h > AULE_MAX_TENSORS) return nullptr;
    DevTensor* t = &g_tensors[h - 1];
    return t->used ? t : nullptr;
}

void free_tensor(DevTensor* t) {
    if (t->used && t->ptr) (void)hipFree(t->ptr);
    *t = DevTensor();
}

// Padded temporary for the host-pointer entry points.
struct Temp {
    float* ptr = nullptr;
    ~Temp() {
        if (ptr) (void)hipFree(ptr);
    }
    bool alloc(size_t rows, uint32_t pitch) {
        if (hipMalloc((void**)&ptr, rows * pitch * sizeof(float)) != hipSuccess) return false;
        return hipMemset(ptr, 0, rows * pitch * sizeof(float)) == hipSuccess;
    }
};

bool upload_rows(float* dst, uint32_t pitch, const float* src, size_t rows, uint32_t d) {
    return hipMemcpy2D(dst, pitch * sizeof(float), src, d * sizeof(float), d * sizeof(float), rows,
                       hipMemcpyHostToDevice) == hipSuccess;
}

bool download_rows(float* dst, const float* src, uint32_t pitch, size_t rows, uint32_t d) {
    return hipMemcpy2D(dst, d * sizeof(float), src, pitch * sizeof(float), d * sizeof(float), rows,
                       hipMemcpyDeviceToHost) == hipSuccess;
}

int run_fwd_f32(const float* q, const float* k, const float* v, float* o, float* lse, uint32_t B, uint32_t Hq,
                uint32_t Hkv, uint32_t Sq, uint32_t Sk, uint32_t Dlogical, uint32_t Dp, int causal, int window = -1) {
    FwdArgs a;
    a.window = (window > 0 && (uint32_t)window < Sq) ? window : -1;  // W >= Sq masks nothing
    a.q = q; a.k = k; a.v = v; a.o = o; a.lse = lse;
    a.B = (int)B; a.Hq = (int)Hq; a.Hkv = (int)Hkv; a.Sq = (int)Sq; a.Sk = (int)Sk; a.D = (int)Dp;
    a.scale = 1.0f / std::sqrt((float)Dlogical);  // attention_pipeline.zig:329
    a.causal = causal != 0;
    a.dtype = aule_hip::kF32;
    int rc = aule_hip::launch_fwd(a, nullptr);
    if (rc != 0) return rc;
    return (int)hipDeviceSynchronize();
}

}  // namespace

namespace aule_hip {
#ifdef AULE_DEBUG_HOOKS
int launch_fwd_pp_timeline(const FwdArgs& a, unsigned long long* dbg, hipStream_t stream);
int launch_fwd_w4_timeline(const FwdArgs& a, unsigned long long* dbg, hipStream_t stream);
#endif
int configure_fwd();
int configure_bwd();
int configure_kernels() {
    int rc = configure_fwd();
    if (rc) return rc;
    return configure_bwd();
}
}  // namespace aule_hip

extern "C" {

int32_t aule_init(void) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_init) return 0;  // idempotent (src/lib.zig:60-63)
    // AULE_BACKEND (src/backends/backend.zig:86-100): "hip" forces the backend this library IS -- a no-op; any value the
    // reference does not know falls through to its auto-detection, i.e. to HIP here; "vulkan" and "cpu" name backends
    // this build does not contain (no multi-backend dispatch, no CPU fallback): fail loudly instead of running something else.
    if (const char* b = getenv("AULE_BACKEND")) {
        if (strcmp(b, "vulkan") == 0 || strcmp(b, "cpu") == 0) {
            set_error("Failed to initialize backend: AULE_BACKEND=%s, but this library contains the HIP (gfx950) backend only", b);
            return -1;
        }
    }
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        set_error("Failed to initialize backend: no HIP device (%s)",
                  e == hipSuccess ? "device count 0" : hipGetErrorString(e));
        return -1;
    }
    int dev = 0;
    if (const char* s = getenv("AULE_HIP_DEVICE")) {
        dev = atoi(s);
        if (dev < 0 || dev >= n) {
            set_error("Failed to initialize backend: AULE_HIP_DEVICE=%d out of range (0..%d)", dev, n - 1);
            return -1;
        }
    } else if (hipGetDevice(&dev) != hipSuccess) {
        dev = 0;
    }
    g_device = dev;
    {
        DeviceGuard g(g_device);
        int rc = ensure_configured();
        if (rc != 0) return -1;
    }
    g_init = true;
    g_err_len = 0;
    return 0;
}

void aule_shutdown(void) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_init) {
        DeviceGuard g(g_device);
        for (auto& t : g_tensors) free_tensor(&t);
    }
    g_init = false;
}

const char* aule_get_error(void) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_err_len == 0) return "No error";
    g_err[g_err_len] = 0;
    return g_err;
}

const char* aule_get_backend_name(void) {
    std::lock_guard<std::mutex> lk(g_mu);
    return g_init ? "HIP/ROCm" : "Not initialized";  // backend.zig:496-502
}

int32_t aule_get_vendor(void) {
    std::lock_guard<std::mutex> lk(g_mu);
    return g_init ? 1 : -1;
}
int32_t aule_get_gpu_vendor(void) { return aule_get_vendor(); }

int32_t aule_is_amd_optimized(void) {
    std::lock_guard<std::mutex> lk(g_mu);
    return g_init ? 1 : -1;
}

int32_t aule_has_fp16(void) {
    std::lock_guard<std::mutex> lk(g_mu);
    return g_init ? 1 : -1;
}

int32_t aule_get_subgroup_size(void) {
    std::lock_guard<std::mutex> lk(g_mu);
    return g_init ? 64 : -1;
}

int32_t aule_get_device_name(uint8_t* buffer, uint32_t buffer_len) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_init) return -1;
    if (buffer == nullptr || buffer_len == 0) return 0;
    hipDeviceProp_t prop;
    const char* name = "HIP Device";
    if (hipGetDeviceProperties(&prop, g_device) == hipSuccess) name = prop.name;
    size_t n = strlen(name);
    if (n > buffer_len - 1) n = buffer_len - 1;
    memcpy(buffer, name, n);
    buffer[n] = 0;
    return (int32_t)n;
}

int32_t aule_set_shader_variant(uint8_t variant) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_init) return -1;
    if (variant != 0) {
        set_error("Shader variant %u not available (the HIP build has one kernel family)", (unsigned)variant);
        return -2;
    }
    g_variant = 0;
    return 0;
}

int32_t aule_get_shader_variant(void) {
    std::lock_guard<std::mutex> lk(g_mu);
    return g_init ? g_variant : -1;
}

int32_t aule_has_shader_variant(uint8_t variant) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_init) return -1;
    return variant == 0 ? 1 : 0;
}

int32_t aule_supports_backward(void) {
    std::lock_guard<std::mutex> lk(g_mu);
    return g_init ? 1 : 0;
}

/* ---------------------------------------------------------------- tensors */
aule_tensor_handle aule_tensor_create(uint32_t b, uint32_t h, uint32_t s, uint32_t d) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_init) {
        set_error("Not initialized");
        return 0;
    }
    int slot = -1;
    for (int i = 0; i < (int)AULE_MAX_TENSORS; ++i)
        if (!g_tensors[i].used) {
            slot = i;
            break;
        }
    if (slot < 0) {
        set_error("Max tensors reached");
        return 0;
    }
    const uint64_t rows = (uint64_t)b * h * s;
    const uint32_t pitch = pad_dim(d);
    DevTensor t;
    t.shape[0] = b; t.shape[1] = h; t.shape[2] = s; t.shape[3] = d;
    t.count = rows * d;
    t.pitch = pitch;
    const size_t bytes = (size_t)rows * pitch * sizeof(float);
    DeviceGuard g(g_device);
    if (bytes > 0) {
        hipError_t e = hipMalloc(&t.ptr, bytes);
        if (e != hipSuccess) {
            set_error("Create tensor failed: %s", hipGetErrorString(e));
            return 0;
        }
        (void)hipMemset(t.ptr, 0, bytes);
    }
    t.used = true;
    g_tensors[slot] = t;
    return (aule_tensor_handle)(slot + 1);
}

aule_tensor_handle aule_tensor_create_u32(uint32_t b, uint32_t h, uint32_t s, uint32_t d) {
    return aule_tensor_create(b, h, s, d);  // src/lib.zig:432-442: u32 aliases fp32 storage
}

void aule_tensor_destroy(aule_tensor_handle handle) {
    std::lock_guard<std::mutex> lk(g_mu);
    DevTensor* t = lookup(handle);
    if (!t) return;
    DeviceGuard g(g_device);
    free_tensor(t);
}

int32_t aule_tensor_upload(aule_tensor_handle handle, const float* data, uint32_t count) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_init) return -1;
    DevTensor* t = lookup(handle);
    if (!t) return -1;
    if ((uint64_t)count != t->count) {  // backend.zig:277
        set_error("Upload failed: size mismatch (tensor has %llu elements, got %u)",
                  (unsigned long long)t->count, count);
        return -3;
    }
    if (count == 0) return 0;
    DeviceGuard g(g_device);
    const size_t rows = (size_t)t->shape[0] * t->shape[1] * t->shape[2];
    if (!upload_rows((float*)t->ptr, t->pitch, data, rows, t->shape[3])) {
        set_error("Upload failed: %s", hipGetErrorString(hipGetLastError()));
        return -3;
    }
    return 0;
}

int32_t aule_tensor_download(aule_tensor_handle handle, float* output, uint32_t count) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_init) return -1;
    DevTensor* t = lookup(handle);
    if (!t) return -1;
    if ((uint64_t)count != t->count) {  // backend.zig:298
        set_error("Download failed: size mismatch (tensor has %llu elements, got %u)",
                  (unsigned long long)t->count, count);
        return -3;
    }
    if (count == 0) return 0;
    DeviceGuard g(g_device);
    const size_t rows = (size_t)t->shape[0] * t->shape[1] * t->shape[2];
    if (!download_rows(output, (const float*)t->ptr, t->pitch, rows, t->shape[3])) {
        set_error("Download failed: %s", hipGetErrorString(hipGetLastError()));
        return -3;
    }
    return 0;
}

int32_t aule_tensor_download_u32(aule_tensor_handle handle, uint32_t* output, uint32_t count) {
    return aule_tensor_download(handle, reinterpret_cast<float*>(output), count);
}

uint32_t aule_tensor_size(aule_tensor_handle handle) {
    std::lock_guard<std::mutex> lk(g_mu);
    DevTensor* t = lookup(handle);
    return t ? (uint32_t)t->count : 0;
}

uint32_t aule_tensor_count(void) {
    std::lock_guard<std::mutex> lk(g_mu);
    uint32_t n = 0;
    for (auto& t : g_tensors) n += t.used ? 1 : 0;
    return n;
}

uint32_t aule_tensor_max(void) { return AULE_MAX_TENSORS; }

void aule_tensor_clear_all(void) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_init) return;
    DeviceGuard g(g_device);
    for (auto& t : g_tensors) free_tensor(&t);
}

/* ------------------------------------------------------- handle-based fwd */
int32_t aule_attention_forward_gpu(aule_tensor_handle qh, aule_tensor_handle kh, aule_tensor_handle vh,
                                   aule_tensor_handle oh, aule_tensor_handle rot_cos,
                                   aule_tensor_handle rot_sin, int32_t causal, int32_t window_size) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_init) return -1;
    DevTensor* q = lookup(qh);
    DevTensor* k = lookup(kh);
    DevTensor* v = lookup(vh);
    DevTensor* o = lookup(oh);
    if (!q || !k || !v || !o) return -1;
    // rot_cos / rot_sin: both or neither; [.., S, D/2] tables, interleaved pairs (shaders/attention_f32.comp:98-111)
    DevTensor* rc_t = nullptr;
    DevTensor* rs_t = nullptr;
    if (rot_cos != 0 || rot_sin != 0) {
        rc_t = lookup(rot_cos);
        rs_t = lookup(rot_sin);
        if (!rc_t || !rs_t) return -1;
    }
    // window_size > 0: sliding window, key j visible to query i only if i - j < window_size (the convention of
    // the kernel the reference runs on ROCm, triton_flash_amd.py:179-183; the Vulkan shaders use others)
    // shape rules of attention_gpu.zig:383-404
    const uint32_t B = q->shape[0], Hq = q->shape[1], Sq = q->shape[2], D = q->shape[3];
    const uint32_t Hkv = k->shape[1], Sk = k->shape[2];
    bool ok = k->shape[0] == B && Hkv != 0 && Hq % Hkv == 0 && k->shape[3] == D;
    ok = ok && v->shape[0] == B && v->shape[1] == Hkv && v->shape[2] == Sk && v->shape[3] == D;
    ok = ok && o->shape[0] == B && o->shape[1] == Hq && o->shape[2] == Sq && o->shape[3] == D;
    if (!ok) {
        set_error("Attention failed: error.ShapeMismatch");
        return -3;
    }
    if (D > 128) {
        set_error("Attention failed: error.HeadDimTooLarge (head_dim %u > 128)", D);
        return -3;
    }
    if (q->count == 0) return 0;
    if (Sk == 0) {
        set_error("Attention failed: error.ShapeMismatch (empty key sequence)");
        return -3;
    }
    DeviceGuard g(g_device);
    const float* qp = (const float*)q->ptr;
    const float* kp = (const float*)k->ptr;
    float* rot = nullptr;   // rotated copies of Q and K (the handle tensors are the caller's and stay untouched)
    if (rc_t) {
        // one table shared by every batch and head, indexed by position: a flat [positions, D/2] buffer however its three
        // leading dimensions spell it -- [1, 1, S, D/2], [1, S, 1, D/2], [S, 1, 1, D/2] ("or similar broadcastable": the
        // reference's attention_gpu.zig does not look at the shape at all) -- with at least max(Sq, Sk) positions.  A genuine
        // [B, H, S', D/2] tensor (more than one leading dimension > 1) is refused: it used to pass a flattened-row count check
        // and was then read across head boundaries.
        const uint32_t need = Sq > Sk ? Sq : Sk;
        auto positions = [](const DevTensor* t, uint32_t& n) {
            int big = 0;
            n = 1;
            for (int i = 0; i < 3; ++i)
                if (t->shape[i] != 1) { ++big; n = t->shape[i]; }
            return big <= 1;
        };
        uint32_t nc = 0, ns = 0;
        if ((D & 1) || rc_t->shape[3] != D / 2 || rs_t->shape[3] != D / 2 || !positions(rc_t, nc) || !positions(rs_t, ns) ||
            nc < need || ns < need || rc_t->pitch != rs_t->pitch) {
            set_error("Attention failed: error.ShapeMismatch (rot_cos / rot_sin must be one [>= seq, head_dim/2] table: "
                      "at most one of the three leading dimensions larger than 1)");
            return -3;
        }
        const size_t nq = (size_t)B * Hq * Sq * q->pitch, nk = (size_t)B * Hkv * Sk * k->pitch;
        if (hipMalloc((void**)&rot, (nq + nk) * sizeof(float)) != hipSuccess) {
            set_error("Attention failed: error.OutOfDeviceMemory");
            return -3;
        }
        // The pass writes columns [0, D) only; the fp32 kernels run at the padded width and need pad = 0
        // (handle tensors are zero-padded on upload), so the copies must not carry allocator garbage there.
        if (q->pitch != D && hipMemset(rot, 0, (nq + nk) * sizeof(float)) != hipSuccess) {
            (void)hipFree(rot);
            set_error("Attention failed: RoPE pass: could not clear the workspace");
            return -3;
        }
        aule_hip::RopeArgs r;
        r.cos = (const float*)rc_t->ptr; r.sin = (const float*)rs_t->ptr; r.table_pitch = (int)rc_t->pitch;
        r.D = (int)D; r.layout = AULE_ROPE_INTERLEAVED; r.inverse = 0; r.pos_offset = 0; r.dtype = aule_hip::kF32;
        r.in = q->ptr; r.out = rot; r.nheads = (long long)B * Hq; r.S = (int)Sq; r.pitch = (int)q->pitch;
        int e = aule_hip::launch_rope(r, nullptr);
        r.in = k->ptr; r.out = rot + nq; r.nheads = (long long)B * Hkv; r.S = (int)Sk; r.pitch = (int)k->pitch;
        if (e == 0) e = aule_hip::launch_rope(r, nullptr);
        if (e != 0) {
            (void)hipFree(rot);
            set_error("Attention failed: RoPE pass: %s", e > 0 ? hipGetErrorString((hipError_t)e) : "unsupported shape");
            return -3;
        }
        qp = rot; kp = rot + nq;
    }
    int rc = run_fwd_f32(qp, kp, (const float*)v->ptr, (float*)o->ptr, nullptr,
                         B, Hq, Hkv, Sq, Sk, D, q->pitch, causal, window_size);
    if (rot) (void)hipFree(rot);   // run_fwd_f32 synchronises
    if (rc != 0) {
        set_error("Attention failed: %s", rc > 0 ? hipGetErrorString((hipError_t)rc) : "unsupported shape");
        return -3;
    }
    return 0;
}

/* ------------------------------------------------------ host-pointer paths */
static int32_t forward_host(const float* query, const float* key, const float* value, float* output, float* lse,
                            uint32_t B, uint32_t H, uint32_t S, uint32_t D, int32_t causal) {
    if (!g_init) {
        set_error("Library not initialized. Call aule_init() first.");
        return -1;
    }
    if (D > 128) {
        set_error("Attention failed: error.HeadDimTooLarge (head_dim %u > 128)", D);
        return -4;
    }
    const size_t rows = (size_t)B * H * S;
    if (rows == 0 || D == 0) return 0;
    const uint32_t Dp = pad_dim(D);
    DeviceGuard g(g_device);
    Temp q, k, v, o, l;
    if (!q.alloc(rows, Dp) || !k.alloc(rows, Dp) || !v.alloc(rows, Dp) || !o.alloc(rows, Dp) ||
        (lse && !l.alloc(rows, 1))) {
        set_error("Create tensor failed: %s", hipGetErrorString(hipGetLastError()));
        return -2;
    }
    if (!upload_rows(q.ptr, Dp, query, rows, D) || !upload_rows(k.ptr, Dp, key, rows, D) ||
        !upload_rows(v.ptr, Dp, value, rows, D)) {
        set_error("Upload failed: %s", hipGetErrorString(hipGetLastError()));
        return -3;
    }
    int rc = run_fwd_f32(q.ptr, k.ptr, v.ptr, o.ptr, lse ? l.ptr : nullptr, B, H, H, S, S, D, Dp, causal);
    if (rc != 0) {
        set_error("Attention failed: %s", rc > 0 ? hipGetErrorString((hipError_t)rc) : "unsupported shape");
        return -4;
    }
    if (!download_rows(output, o.ptr, Dp, rows, D) ||
        (lse && hipMemcpy(lse, l.ptr, rows * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess)) {
        set_error("Download failed: %s", hipGetErrorString(hipGetLastError()));
        return -5;
    }
    return 0;
}

int32_t aule_attention_forward(const float* query, const float* key, const float* value, float* output,
                               uint32_t B, uint32_t H, uint32_t S, uint32_t D, int32_t causal) {
    std::lock_guard<std::mutex> lk(g_mu);
    return forward_host(query, key, value, output, nullptr, B, H, S, D, causal);
}

int32_t aule_attention_forward_with_lse(const float* query, const float* key, const float* value, float* output,
                                        float* lse, uint32_t B, uint32_t H, uint32_t S, uint32_t D,
                                        int32_t causal) {
    std::lock_guard<std::mutex> lk(g_mu);
    return forward_host(query, key, value, output, lse, B, H, S, D, causal);
}

int32_t aule_attention_backward(const float* query, const float* key, const float* value, const float* output,
                                const float* grad_output, const float* lse, float* grad_query, float* grad_key,
                                float* grad_value, uint32_t B, uint32_t H, uint32_t S, uint32_t D,
                                int32_t causal) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_init) {
        set_error("Library not initialized. Call aule_init() first.");
        return -1;
    }
    if (D > 128) {
        set_error("Backward failed: error.HeadDimTooLarge (head_dim %u > 128)", D);
        return -4;
    }
    const size_t rows = (size_t)B * H * S;
    if (rows == 0 || D == 0) return 0;
    const uint32_t Dp = pad_dim(D);
    DeviceGuard g(g_device);
    Temp q, k, v, o, go, l, dq, dk, dv, ws;
    const uint64_t wsb = aule_hip::bwd_workspace_bytes((int)B, (int)H, (int)H, (int)S, (int)S, (int)Dp, causal != 0, aule_hip::kF32);
    if (!q.alloc(rows, Dp) || !k.alloc(rows, Dp) || !v.alloc(rows, Dp) || !o.alloc(rows, Dp) ||
        !go.alloc(rows, Dp) || !l.alloc(rows, 1) || !dq.alloc(rows, Dp) || !dk.alloc(rows, Dp) ||
        !dv.alloc(rows, Dp) || !ws.alloc((wsb + 3) / 4, 1)) {
        set_error("Create tensor failed: %s", hipGetErrorString(hipGetLastError()));
        return -2;
    }
    if (!upload_rows(q.ptr, Dp, query, rows, D) || !upload_rows(k.ptr, Dp, key, rows, D) ||
        !upload_rows(v.ptr, Dp, value, rows, D) || !upload_rows(o.ptr, Dp, output, rows, D) ||
        !upload_rows(go.ptr, Dp, grad_output, rows, D) ||
        hipMemcpy(l.ptr, lse, rows * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) {
        set_error("Upload failed: %s", hipGetErrorString(hipGetLastError()));
        return -3;
    }
    BwdArgs a;
    a.q = q.ptr; a.k = k.ptr; a.v = v.ptr; a.o = o.ptr; a.dout = go.ptr; a.lse = l.ptr;
    a.dq = dq.ptr; a.dk = dk.ptr; a.dv = dv.ptr; a.delta = ws.ptr;
    a.B = (int)B; a.Hq = (int)H; a.Hkv = (int)H; a.Sq = (int)S; a.Sk = (int)S; a.D = (int)Dp;
    a.scale = 1.0f / std::sqrt((float)D);
    a.causal = causal != 0;
    a.dtype = aule_hip::kF32;
    int rc = aule_hip::launch_bwd(a, nullptr);
    if (rc == 0) rc = (int)hipDeviceSynchronize();
    if (rc != 0) {
        set_error("Backward failed: %s", rc > 0 ? hipGetErrorString((hipError_t)rc) : "unsupported shape");
        return -4;
    }
    if (!download_rows(grad_query, dq.ptr, Dp, rows, D) || !download_rows(grad_key, dk.ptr, Dp, rows, D) ||
        !download_rows(grad_value, dv.ptr, Dp, rows, D)) {
        set_error("Download failed: %s", hipGetErrorString(hipGetLastError()));
        return -5;
    }
    return 0;
}

/* ------------------------------------------------------ out-of-scope stubs */
int32_t aule_attention_forward_paged(aule_tensor_handle, aule_tensor_handle, aule_tensor_handle,
                                     aule_tensor_handle, aule_tensor_handle, aule_tensor_handle, int32_t,
                                     int32_t) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_init) return -1;
    set_error("PagedAttention failed: not supported by the HIP backend");
    return -3;
}

int32_t aule_spatial_sort(aule_tensor_handle, aule_tensor_handle, aule_tensor_handle, uint32_t) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_init) return -1;
    set_error("Spatial sort failed: not supported by the HIP backend");
    return -3;
}

int32_t aule_attention_forward_gravity(aule_tensor_handle, aule_tensor_handle, aule_tensor_handle,
                                       aule_tensor_handle, aule_tensor_handle, aule_tensor_handle,
                                       aule_tensor_handle, int32_t, uint32_t, int32_t) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_init) return -1;
    set_error("Gravity Attention failed: not supported by the HIP backend");
    return -3;
}

/* ------------------------------------------------------------ _ex entries */
static int check_common(int32_t dtype, uint32_t B, uint32_t Hq, uint32_t Hkv, uint32_t Sq, uint32_t Sk,
                        uint32_t D, int32_t window, int32_t causal) {
    if (causal < 0 || causal > AULE_CAUSAL_BOTTOM_RIGHT) {
        set_error("Attention failed: unknown causal mode %d (0 none, 1 top-left, 2 bottom-right)", causal);
        return -3;
    }
    if (causal == AULE_CAUSAL_BOTTOM_RIGHT && Sk < Sq) {
        set_error("Attention failed: bottom-right causal alignment needs seq_k (%u) >= seq_q (%u)", Sk, Sq);
        return -3;
    }
    if (dtype < 0 || dtype > 2) {
        set_error("Attention failed: unknown dtype %d", dtype);
        return -3;
    }
    if (D != 32 && D != 64 && D != 128) {
        set_error("Attention failed: head_dim %u unsupported (32, 64 or 128; pad to the next size)", D);
        return -3;
    }
    if (Hkv == 0 || Hq % Hkv != 0) {
        set_error("Attention failed: heads_q (%u) must be divisible by heads_kv (%u)", Hq, Hkv);
        return -3;
    }
    (void)window;  // any value is accepted: <= 0 means full attention
    // per-head K/V/Q slabs are addressed through 32-bit buffer descriptors (raw SRD, byte offsets)
    if ((uint64_t)B * Hq * Sq * D >= (1ull << 40) || (uint64_t)Sq * D * 4 >= (1ull << 31) ||
        (uint64_t)Sk * D * 4 >= (1ull << 31)) {
        set_error("Attention failed: problem too large");
        return -3;
    }
    return 0;
}

// One query at the bottom-right position sees every key: with no window the causal mask masks nothing, and the
// problem is the non-causal one (which has the faster short-query paths).
static void drop_trivial_causal(int& causal, int& coff, int Sq, int window) {
    if (causal && Sq == 1 && coff > 0 && window <= 0) {
        causal = 0;
        coff = 0;
    }
}

// roctx ranges around the launches (the reference has no tracing at all), so that
// `rocprofv3 --marker-trace` shows "aule.forward" / "aule.backward" / "aule.paged_decode" / "aule.rope" next to the kernels.
// Opt-in (AULE_ROCTX=1) and loaded lazily with dlopen: the library keeps its single link dependency (libamdhip64).
struct RoctxRange {
    using PushFn = int (*)(const char*);
    using PopFn = int (*)();
    static PushFn push_fn() {
        static const PushFn fn = [] {
            const char* e = getenv("AULE_ROCTX");
            if (e == nullptr || e[0] != '1') return (PushFn) nullptr;
            // rocprofv3 intercepts the rocprofiler-sdk flavour; libroctx64 is the legacy (roctracer) one
            void* h = nullptr;
            for (const char* name : {"librocprofiler-sdk-roctx.so", "/opt/rocm/lib/librocprofiler-sdk-roctx.so", "libroctx64.so",
                                     "/opt/rocm/lib/libroctx64.so"}) {
                h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
                if (h != nullptr) break;
            }
            if (h == nullptr) return (PushFn) nullptr;
            // both or neither: a push whose pop did not resolve would leave a range open on every API call
            const PopFn pop = reinterpret_cast<PopFn>(dlsym(h, "roctxRangePop"));
            const PushFn push = reinterpret_cast<PushFn>(dlsym(h, "roctxRangePushA"));
            if (pop == nullptr || push == nullptr) return (PushFn) nullptr;
            pop_slot() = pop;
            return push;
        }();
        return fn;
    }
    static PopFn& pop_slot() {
        static PopFn fn = nullptr;
        return fn;
    }
    bool on = false;
    explicit RoctxRange(const char* name) {
        if (PushFn f = push_fn()) {
            f(name);
            on = pop_slot() != nullptr;
        }
    }
    ~RoctxRange() {
        if (on) pop_slot()();
    }
    RoctxRange(const RoctxRange&) = delete;
    RoctxRange& operator=(const RoctxRange&) = delete;
};

static float resolve_scale(float scale, uint32_t D) {
    if (scale == 0.0f || std::isnan(scale)) return 1.0f / std::sqrt((float)D);
    return scale;
}

// Descriptor -> launch arguments (shared by the plain and the fused-rotation entry points; no device work).
static void fill_fwd_args(const aule_attn_desc* d, FwdArgs& a) {
    a.q = d->q; a.k = d->k; a.v = d->v; a.o = d->out; a.lse = d->lse;
    a.B = (int)d->batch; a.Hq = (int)d->heads_q; a.Hkv = (int)d->heads_kv;
    a.Sq = (int)d->seq_q; a.Sk = (int)d->seq_k; a.D = (int)d->head_dim;
    a.scale = resolve_scale(d->scale, d->head_dim);
    a.causal = d->causal != 0;
    a.coff = d->causal == AULE_CAUSAL_BOTTOM_RIGHT ? (int)d->seq_k - (int)d->seq_q : 0;
    a.dtype = d->dtype; a.device = d->device;
    // W >= Sq + coff masks nothing (the last query sits at position Sq - 1 + coff)
    a.window = (d->window_size > 0 && (uint32_t)d->window_size < d->seq_q + (uint32_t)a.coff) ? d->window_size : -1;
    drop_trivial_causal(a.causal, a.coff, a.Sq, a.window);
    a.ws = d->workspace; a.ws_bytes = d->workspace ? d->workspace_bytes : 0;
}

static bool fill_rope_args(const aule_attn_rope* r, uint32_t head_dim, FwdArgs& a) {
    if (r == nullptr || r->struct_size != sizeof(aule_attn_rope) || r->layout != AULE_ROPE_HALF) return false;
    if (r->table_len >= (1u << 30) || r->table_pitch >= (1u << 20) || r->q_pos_offset >= (1u << 30)) return false;
    a.rope_cos = r->cos; a.rope_sin = r->sin;
    a.rope_rows = (int)r->table_len;
    a.rope_pitch = (int)(r->table_pitch ? r->table_pitch : head_dim / 2);
    a.rope_pos = (int)r->q_pos_offset;
    return true;
}

static int32_t forward_impl(const aule_attn_desc* d, const aule_attn_rope* rope) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_init) {
        set_error("Library not initialized. Call aule_init() first.");
        return -1;
    }
    if (d == nullptr || d->struct_size != sizeof(aule_attn_desc)) {
        set_error("Attention failed: bad descriptor (struct_size mismatch)");
        return -3;
    }
    int rc = check_common(d->dtype, d->batch, d->heads_q, d->heads_kv, d->seq_q, d->seq_k, d->head_dim,
                          d->window_size, d->causal);
    if (rc) return rc;
    if ((uint64_t)d->batch * d->heads_q * d->seq_q == 0) return 0;  // empty output
    if (d->seq_k == 0) {
        set_error("Attention failed: empty key sequence");
        return -3;
    }
    if (!d->q || !d->k || !d->v || !d->out) {
        set_error("Attention failed: null tensor pointer");
        return -3;
    }
    DeviceGuard g(d->device);
    rc = ensure_configured();
    if (rc) return rc;
    FwdArgs a;
    fill_fwd_args(d, a);
    if (rope != nullptr) {
        if (!fill_rope_args(rope, d->head_dim, a) || !aule_hip::fwd_rope_fusable(a)) {
            set_error("Attention failed: the query rotation is not fused for this configuration "
                      "(aule_attention_forward_rope_fusable() == 0): rotate Q with aule_rope_ex() and call aule_attention_forward_ex()");
            return -3;
        }
    }
    rc = aule_hip::launch_fwd(a, (hipStream_t)d->stream);
    if (rc != 0) {
        set_error("Attention failed: %s", rc > 0 ? hipGetErrorString((hipError_t)rc) : "unsupported configuration");
        return -4;
    }
    return 0;
}

int32_t aule_attention_forward_ex(const aule_attn_desc* d) {
    RoctxRange range("aule.forward");
    return forward_impl(d, nullptr);
}

int32_t aule_attention_forward_rope_ex(const aule_attn_desc* d, const aule_attn_rope* rope) {
    RoctxRange range("aule.forward_rope");
    if (rope == nullptr) {
        std::lock_guard<std::mutex> lk(g_mu);
        set_error("Attention failed: null rotation descriptor");
        return -3;
    }
    return forward_impl(d, rope);
}

int32_t aule_attention_forward_rope_fusable(const aule_attn_desc* d, const aule_attn_rope* rope) {
    if (d == nullptr || d->struct_size != sizeof(aule_attn_desc) || rope == nullptr) return 0;
    if (d->dtype < 0 || d->dtype > 2 || d->heads_kv == 0 || d->heads_q % d->heads_kv != 0) return 0;
    if (d->causal < 0 || d->causal > AULE_CAUSAL_BOTTOM_RIGHT || (d->causal == AULE_CAUSAL_BOTTOM_RIGHT && d->seq_k < d->seq_q)) return 0;
    if ((uint64_t)d->batch * d->heads_q * d->seq_q == 0 || d->seq_k == 0) return 0;
    if (d->batch >= (1u << 24) || d->heads_q >= (1u << 24) || d->seq_q >= (1u << 30) || d->seq_k >= (1u << 30)) return 0;
    FwdArgs a;
    fill_fwd_args(d, a);
    return fill_rope_args(rope, d->head_dim, a) && aule_hip::fwd_rope_fusable(a) ? 1 : 0;
}

int32_t aule_attention_paged_decode_ex(const aule_paged_desc* d) {
    RoctxRange range("aule.paged_decode");
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_init) {
        set_error("Library not initialized. Call aule_init() first.");
        return -1;
    }
    if (d == nullptr || d->struct_size != sizeof(aule_paged_desc)) {
        set_error("Paged attention failed: bad descriptor (struct_size mismatch)");
        return -3;
    }
    if (d->dtype != AULE_DTYPE_F16 && d->dtype != AULE_DTYPE_BF16) {
        set_error("Paged attention failed: dtype must be fp16 or bf16");
        return -3;
    }
    if (d->head_dim != 32 && d->head_dim != 64 && d->head_dim != 128) {
        set_error("Paged attention failed: head_dim %u unsupported (32, 64 or 128)", d->head_dim);
        return -3;
    }
    if (d->heads_kv == 0 || d->heads_q % d->heads_kv != 0) {
        set_error("Paged attention failed: heads_q (%u) must be divisible by heads_kv (%u)", d->heads_q, d->heads_kv);
        return -3;
    }
    if (d->block_size == 0 || d->max_blocks == 0 || (uint64_t)d->block_size * d->max_blocks >= (1ull << 30)) {
        set_error("Paged attention failed: bad block_size / max_blocks");
        return -3;
    }
    if ((uint64_t)d->batch * d->heads_q == 0) return 0;
    if (!d->q || !d->k_cache || !d->v_cache || !d->block_tables || !d->context_lens || !d->out) {
        set_error("Paged attention failed: null tensor pointer");
        return -3;
    }
    DeviceGuard g(d->device);
    int rc = ensure_configured();
    if (rc) return rc;
    aule_hip::PagedArgs a;
    a.q = d->q; a.k_cache = d->k_cache; a.v_cache = d->v_cache; a.out = d->out;
    a.block_tables = d->block_tables; a.context_lens = d->context_lens;
    a.B = (int)d->batch; a.Hq = (int)d->heads_q; a.Hkv = (int)d->heads_kv; a.D = (int)d->head_dim;
    a.block_size = (int)d->block_size; a.max_blocks = (int)d->max_blocks;
    a.scale = resolve_scale(d->scale, d->head_dim);
    a.window = d->window_size;
    a.dtype = d->dtype;
    a.ws = d->workspace; a.ws_bytes = d->workspace ? d->workspace_bytes : 0;
    rc = aule_hip::launch_paged_decode(a, (hipStream_t)d->stream);
    if (rc != 0) {
        set_error("Paged attention failed: %s", rc > 0 ? hipGetErrorString((hipError_t)rc) : "unsupported configuration");
        return -4;
    }
    return 0;
}

int32_t aule_rope_ex(const aule_rope_desc* d) {
    RoctxRange range("aule.rope");
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_init) {
        set_error("Library not initialized. Call aule_init() first.");
        return -1;
    }
    if (d == nullptr || d->struct_size != sizeof(aule_rope_desc)) {
        set_error("RoPE failed: bad descriptor (struct_size mismatch)");
        return -3;
    }
    if (d->dtype < 0 || d->dtype > 2) {
        set_error("RoPE failed: unknown dtype %d", d->dtype);
        return -3;
    }
    if (d->head_dim == 0 || (d->head_dim & 1) || d->row_pitch < d->head_dim) {
        set_error("RoPE failed: head_dim (%u) must be even and <= row_pitch (%u)", d->head_dim, d->row_pitch);
        return -3;
    }
    if (d->layout != AULE_ROPE_HALF && d->layout != AULE_ROPE_INTERLEAVED) {
        set_error("RoPE failed: unknown layout %d", d->layout);
        return -3;
    }
    if ((uint64_t)d->seq + d->pos_offset > d->table_len) {
        set_error("RoPE failed: table too short (%u rows < seq %u + pos_offset %u)", d->table_len, d->seq, d->pos_offset);
        return -3;
    }
    if (d->table_pitch != 0 && d->table_pitch < d->head_dim / 2) {
        set_error("RoPE failed: table_pitch (%u) < head_dim/2", d->table_pitch);
        return -3;
    }
    if (d->rows_bh == 0 || d->seq == 0) return 0;
    if (d->rows_bh * d->seq >= (1ull << 40) || d->seq >= (1u << 30)) {
        set_error("RoPE failed: problem too large");
        return -3;
    }
    if (!d->in || !d->out || !d->cos || !d->sin) {
        set_error("RoPE failed: null pointer");
        return -3;
    }
    DeviceGuard g(d->device);
    aule_hip::RopeArgs r;
    r.in = d->in; r.out = d->out; r.cos = d->cos; r.sin = d->sin;
    r.nheads = (long long)d->rows_bh; r.S = (int)d->seq; r.D = (int)d->head_dim; r.pitch = (int)d->row_pitch;
    r.layout = d->layout; r.inverse = d->inverse != 0; r.pos_offset = (int)d->pos_offset; r.dtype = d->dtype;
    r.table_pitch = (int)d->table_pitch;
    const int rc = aule_hip::launch_rope(r, (hipStream_t)d->stream);
    if (rc != 0) {
        set_error("RoPE failed: %s", rc > 0 ? hipGetErrorString((hipError_t)rc) : "unsupported configuration");
        return -4;
    }
    return 0;
}

uint64_t aule_attention_backward_workspace_size(const aule_attn_bwd_desc* d) {
    if (d == nullptr) return 0;
    // (a window that masks something -- the rule of aule_attention_backward_ex -- keeps the call on the recompute pair: no dS workspace then)
    const int coff = d->causal == AULE_CAUSAL_BOTTOM_RIGHT ? (int)d->seq_k - (int)d->seq_q : 0;
    const bool windowed = d->window_size > 0 && (long long)d->window_size < (long long)d->seq_q + coff;
    return aule_hip::bwd_workspace_bytes((int)d->batch, (int)d->heads_q, (int)d->heads_kv, (int)d->seq_q, (int)d->seq_k,
                                         (int)d->head_dim, d->causal != 0, d->dtype, d->device, windowed);
}

int32_t aule_attention_backward_ex(const aule_attn_bwd_desc* d) {
    RoctxRange range("aule.backward");
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_init) {
        set_error("Library not initialized. Call aule_init() first.");
        return -1;
    }
    if (d == nullptr || d->struct_size != sizeof(aule_attn_bwd_desc)) {
        set_error("Backward failed: bad descriptor (struct_size mismatch)");
        return -3;
    }
    int rc = check_common(d->dtype, d->batch, d->heads_q, d->heads_kv, d->seq_q, d->seq_k, d->head_dim,
                          d->window_size, d->causal);
    if (rc) return rc;
    if ((uint64_t)d->batch * d->heads_q * d->seq_q == 0 && (uint64_t)d->batch * d->heads_kv * d->seq_k == 0)
        return 0;
    if (d->seq_k == 0 || d->seq_q == 0) {
        set_error("Backward failed: empty sequence");
        return -3;
    }
    if (!d->q || !d->k || !d->v || !d->out || !d->dout || !d->lse || !d->dq || !d->dk || !d->dv) {
        set_error("Backward failed: null tensor pointer");
        return -3;
    }
    // (aule_attention_backward_workspace_size() also asks for the dS workspace of the 5-matmul backward; a smaller buffer that
    // still holds delta / L' / the partials is accepted and runs the recompute pair)
    const uint64_t need = aule_hip::bwd_workspace_min_bytes((int)d->batch, (int)d->heads_q, (int)d->heads_kv, (int)d->seq_q,
                                                            (int)d->seq_k, (int)d->head_dim, d->causal != 0, d->dtype, d->device);
    if (!d->workspace || d->workspace_bytes < need) {
        set_error("Backward failed: workspace too small (%llu < %llu bytes)",
                  (unsigned long long)d->workspace_bytes, (unsigned long long)need);
        return -3;
    }
    DeviceGuard g(d->device);
    rc = ensure_configured();
    if (rc) return rc;
    BwdArgs a;
    a.q = d->q; a.k = d->k; a.v = d->v; a.o = d->out; a.dout = d->dout; a.lse = d->lse;
    a.dq = d->dq; a.dk = d->dk; a.dv = d->dv; a.delta = (float*)d->workspace;
    a.ws_bytes = d->workspace_bytes;
    a.device = d->device;
    a.B = (int)d->batch; a.Hq = (int)d->heads_q; a.Hkv = (int)d->heads_kv;
    a.Sq = (int)d->seq_q; a.Sk = (int)d->seq_k; a.D = (int)d->head_dim;
    a.scale = resolve_scale(d->scale, d->head_dim);
    a.causal = d->causal != 0;
    a.coff = d->causal == AULE_CAUSAL_BOTTOM_RIGHT ? (int)d->seq_k - (int)d->seq_q : 0;
    a.dtype = d->dtype;
    a.window = (d->window_size > 0 && (uint32_t)d->window_size < d->seq_q + (uint32_t)a.coff) ? d->window_size : -1;
    drop_trivial_causal(a.causal, a.coff, a.Sq, a.window);
    rc = aule_hip::launch_bwd(a, (hipStream_t)d->stream);
    if (rc != 0) {
        set_error("Backward failed: %s", rc > 0 ? hipGetErrorString((hipError_t)rc) : "unsupported configuration");
        return -4;
    }
    return 0;
}

// ---- direct peer exchange (include/aule.h; consumer: aule/dist.py, transport="peer")
static_assert(sizeof(aule_ipc_handle) == sizeof(hipIpcMemHandle_t), "aule_ipc_handle must carry a hipIpcMemHandle_t");

int32_t aule_peer_alloc(int32_t device, uint64_t bytes, void** ptr, aule_ipc_handle* handle) {
    if (ptr == nullptr || handle == nullptr || bytes == 0) { set_error("aule_peer_alloc: bad argument"); return -1; }
    DeviceGuard g(device);
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess) { set_error("aule_peer_alloc: hipMalloc(%llu): %s", (unsigned long long)bytes, hipGetErrorString(e)); return -2; }
    hipIpcMemHandle_t h;
    e = hipIpcGetMemHandle(&h, p);
    if (e != hipSuccess) {
        set_error("aule_peer_alloc: hipIpcGetMemHandle: %s (HSA_ENABLE_IPC_MODE_LEGACY=0 set?)", hipGetErrorString(e));
        (void)hipFree(p);
        return -4;
    }
    std::memcpy(handle->bytes, &h, sizeof(h));
    *ptr = p;
    return 0;
}

int32_t aule_peer_free(int32_t device, void* ptr) {
    if (ptr == nullptr) return 0;
    DeviceGuard g(device);
    const hipError_t e = hipFree(ptr);
    if (e != hipSuccess) { set_error("aule_peer_free: %s", hipGetErrorString(e)); return -4; }
    return 0;
}

int32_t aule_peer_open(int32_t device, const aule_ipc_handle* handle, void** ptr) {
    if (ptr == nullptr || handle == nullptr) { set_error("aule_peer_open: bad argument"); return -1; }
    DeviceGuard g(device);
    hipIpcMemHandle_t h;
    std::memcpy(&h, handle->bytes, sizeof(h));
    void* p = nullptr;
    const hipError_t e = hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess);
    if (e != hipSuccess) { set_error("aule_peer_open: hipIpcOpenMemHandle: %s", hipGetErrorString(e)); return -4; }
    *ptr = p;
    return 0;
}

int32_t aule_peer_close(int32_t device, void* ptr) {
    if (ptr == nullptr) return 0;
    DeviceGuard g(device);
    const hipError_t e = hipIpcCloseMemHandle(ptr);
    if (e != hipSuccess) { set_error("aule_peer_close: %s", hipGetErrorString(e)); return -4; }
    return 0;
}

int32_t aule_peer_copy_async(int32_t device, void* dst, const void* src, uint64_t bytes, void* stream) {
    if (bytes == 0) return 0;
    if (dst == nullptr || src == nullptr) { set_error("aule_peer_copy_async: bad argument"); return -1; }
    DeviceGuard g(device);
    const hipError_t e = hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream);
    if (e != hipSuccess) { set_error("aule_peer_copy_async: %s", hipGetErrorString(e)); return -4; }
    return 0;
}

const char* aule_hip_build_info(void) { return "aule-hip gfx950 abi2"; }   // abi2: workspace fields in the fwd / paged descriptors

uint64_t aule_attention_forward_workspace_size(const aule_attn_desc* d) {
    if (d == nullptr || d->struct_size != sizeof(aule_attn_desc)) return 0;
    if (d->dtype < 0 || d->dtype > 2 || d->causal < 0 || d->causal > AULE_CAUSAL_BOTTOM_RIGHT) return 0;
    if (d->heads_kv == 0 || d->heads_q % d->heads_kv != 0 || d->seq_k == 0) return 0;
    if ((uint64_t)d->batch * d->heads_q * d->seq_q == 0) return 0;
    if (d->head_dim != 32 && d->head_dim != 64 && d->head_dim != 128) return 0;
    if (d->causal == AULE_CAUSAL_BOTTOM_RIGHT && d->seq_k < d->seq_q) return 0;
    FwdArgs a;
    a.q = a.k = a.v = nullptr; a.o = nullptr; a.lse = nullptr;
    a.B = (int)d->batch; a.Hq = (int)d->heads_q; a.Hkv = (int)d->heads_kv;
    a.Sq = (int)d->seq_q; a.Sk = (int)d->seq_k; a.D = (int)d->head_dim;
    a.scale = 1.0f;
    a.causal = d->causal != 0;
    a.coff = d->causal == AULE_CAUSAL_BOTTOM_RIGHT ? a.Sk - a.Sq : 0;
    a.dtype = d->dtype; a.device = d->device;
    a.window = (d->window_size > 0 && (uint32_t)d->window_size < d->seq_q + (uint32_t)a.coff) ? d->window_size : -1;
    drop_trivial_causal(a.causal, a.coff, a.Sq, a.window);
    return aule_hip::fwd_workspace_bytes(a);
}

uint64_t aule_attention_paged_decode_workspace_size(const aule_paged_desc* d) {
    if (d == nullptr || d->struct_size != sizeof(aule_paged_desc)) return 0;
    if (d->dtype != AULE_DTYPE_F16 && d->dtype != AULE_DTYPE_BF16) return 0;
    if (d->head_dim != 32 && d->head_dim != 64 && d->head_dim != 128) return 0;
    if (d->heads_kv == 0 || d->heads_q % d->heads_kv != 0) return 0;
    if (d->block_size == 0 || d->max_blocks == 0 || (uint64_t)d->block_size * d->max_blocks >= (1ull << 30)) return 0;
    if ((uint64_t)d->batch * d->heads_q == 0) return 0;
    aule_hip::PagedArgs a;
    a.q = a.k_cache = a.v_cache = nullptr; a.out = nullptr; a.block_tables = nullptr; a.context_lens = nullptr;
    a.B = (int)d->batch; a.Hq = (int)d->heads_q; a.Hkv = (int)d->heads_kv; a.D = (int)d->head_dim;
    a.block_size = (int)d->block_size; a.max_blocks = (int)d->max_blocks;
    a.scale = 1.0f; a.window = d->window_size; a.dtype = d->dtype;
    return aule_hip::paged_workspace_bytes(a);
}

#ifdef AULE_DEBUG_HOOKS
/* The timeline hooks exist only in the debug library (`make dbg` -> build/variants/libaule_dbg.so, -DAULE_DEBUG_HOOKS):
 * they launch instrumented kernel instances on caller-supplied pointers and are not part of the product libaule.so. */
/* Debug hook (not part of the drop-in ABI): aule_attention_backward_ex with the dK/dV kernel's timeline build --
 * per-phase s_memtime stamps of its workgroup 0 into `stamps` (device pointer, 8 * 384 uint64; bf16 D128 causal only,
 * otherwise the ordinary kernels run and nothing is written).  Used by tools/timeline_bwd.py. */
static int32_t backward_timeline(const aule_attn_bwd_desc* d, unsigned long long* stamps, bool dq) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_init || d == nullptr || d->struct_size != sizeof(aule_attn_bwd_desc) || stamps == nullptr) return -1;
    if (d->dtype != AULE_DTYPE_BF16 || (d->head_dim != 128 && !(d->head_dim == 64 && !dq)) || d->heads_kv == 0 || d->heads_q % d->heads_kv != 0) return -3;   // (D = 64: the dK/dV timeline only)
    BwdArgs a;
    a.q = d->q; a.k = d->k; a.v = d->v; a.o = d->out; a.dout = d->dout; a.lse = d->lse;
    a.dq = d->dq; a.dk = d->dk; a.dv = d->dv; a.delta = (float*)d->workspace;
    a.B = (int)d->batch; a.Hq = (int)d->heads_q; a.Hkv = (int)d->heads_kv;
    a.Sq = (int)d->seq_q; a.Sk = (int)d->seq_k; a.D = (int)d->head_dim;
    a.scale = resolve_scale(d->scale, d->head_dim);
    a.causal = d->causal != 0;
    a.dtype = d->dtype;
    if (dq) a.dbg_dq = stamps; else a.dbg = stamps;
    return aule_hip::launch_bwd(a, (hipStream_t)d->stream);
}
int32_t aule_hip_debug_backward_timeline(const aule_attn_bwd_desc* d, unsigned long long* stamps) {
    return backward_timeline(d, stamps, false);
}
/* ... the same for the dQ kernel (8 stamps per tile). */
int32_t aule_hip_debug_backward_timeline_dq(const aule_attn_bwd_desc* d, unsigned long long* stamps) {
    return backward_timeline(d, stamps, true);
}
#endif  // AULE_DEBUG_HOOKS

/* Debug hook (not part of the drop-in ABI): the forward kernel aule_attention_forward_ex would launch for `d`
 * -- 0 fp32, 1 ping-pong, 4 split-KV, 5 ping-pong with packed rows + KV splits, 6 persistent tile stream; -3 for a bad descriptor.  Pure host logic: no
 * device, no aule_init() needed.  Used by the tests to pin which kernel a shape exercises. */
int32_t aule_hip_debug_forward_route(const aule_attn_desc* d) {
    if (d == nullptr || d->struct_size != sizeof(aule_attn_desc)) return -3;
    if (d->causal < 0 || d->causal > AULE_CAUSAL_BOTTOM_RIGHT) return -3;
    FwdArgs a;
    a.B = (int)d->batch; a.Hq = (int)d->heads_q; a.Hkv = (int)d->heads_kv;
    a.Sq = (int)d->seq_q; a.Sk = (int)d->seq_k; a.D = (int)d->head_dim;
    if (a.Hkv <= 0 || a.Hq % a.Hkv != 0) return -3;
    a.causal = d->causal != 0;
    a.coff = d->causal == AULE_CAUSAL_BOTTOM_RIGHT ? a.Sk - a.Sq : 0;
    a.dtype = d->dtype; a.device = d->device;
    a.scale = resolve_scale(d->scale, d->head_dim);   // (the sign of the scale picks the kernel: negative scales stay off route 8)
    a.window = (d->window_size > 0 && (uint32_t)d->window_size < d->seq_q + (uint32_t)a.coff) ? d->window_size : -1;
    drop_trivial_causal(a.causal, a.coff, a.Sq, a.window);
    return aule_hip::fwd_route(a);
}

/* Debug hook: what the most recent backward launch of this process ran (bit mask, include/aule.h). */
int32_t aule_hip_debug_last_backward_route(void) { return aule_hip::bwd_last_route(); }

int32_t aule_hip_debug_forward_split_plan(const aule_attn_desc* d, int32_t* out, int32_t cap) {
    if (d == nullptr || d->struct_size != sizeof(aule_attn_desc)) return -3;
    if (d->causal < 0 || d->causal > AULE_CAUSAL_BOTTOM_RIGHT) return -3;
    FwdArgs a;
    a.B = (int)d->batch; a.Hq = (int)d->heads_q; a.Hkv = (int)d->heads_kv;
    a.Sq = (int)d->seq_q; a.Sk = (int)d->seq_k; a.D = (int)d->head_dim;
    if (a.Hkv <= 0 || a.Hq % a.Hkv != 0) return -3;
    a.causal = d->causal != 0;
    a.coff = d->causal == AULE_CAUSAL_BOTTOM_RIGHT ? a.Sk - a.Sq : 0;
    a.dtype = d->dtype; a.device = d->device;
    a.scale = resolve_scale(d->scale, d->head_dim);
    a.window = (d->window_size > 0 && (uint32_t)d->window_size < d->seq_q + (uint32_t)a.coff) ? d->window_size : -1;
    drop_trivial_causal(a.causal, a.coff, a.Sq, a.window);
    if (aule_hip::fwd_route(a) != 7) return 0;
    return aule_hip::fwd_split_plan_dump(a, out, cap);
}

int32_t aule_hip_debug_work_order(int32_t ranked, int32_t bid, int32_t batch, int32_t heads_q, int32_t heads_kv, int32_t nblk, int32_t flag, int32_t* out4) {
    if (out4 == nullptr || batch <= 0 || heads_kv <= 0 || heads_q <= 0 || heads_q % heads_kv != 0 || nblk <= 0) return -3;
    if (bid < 0 || (int64_t)bid >= (int64_t)batch * heads_q * nblk) return -3;
    int o[4];
    aule_hip::work_order_dump(ranked, bid, batch, heads_q, heads_kv, nblk, flag, o);
    for (int i = 0; i < 4; ++i) out4[i] = o[i];
    return 0;
}

#ifdef AULE_DEBUG_HOOKS
/* Debug hook (debug library only): bf16 D=128 forward with per-phase s_memtime stamps of workgroup 0 written to
 * `stamps` (device pointer; 8 * 256 uint64 for the ping-pong kernel, 8 * 2048 with AULE_TL=ps for the tile stream).
 * Used by tools/timeline.py / tools/timeline_w4.py. */
int32_t aule_hip_debug_forward_timeline(const aule_attn_desc* d, unsigned long long* stamps) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_init || d == nullptr || d->struct_size != sizeof(aule_attn_desc) || stamps == nullptr) return -1;
    if (d->dtype != AULE_DTYPE_BF16 || (d->head_dim != 128 && d->head_dim != 64) || d->heads_kv == 0 || d->heads_q % d->heads_kv != 0) return -3;
    FwdArgs a;
    a.q = d->q; a.k = d->k; a.v = d->v; a.o = d->out; a.lse = d->lse;
    a.B = (int)d->batch; a.Hq = (int)d->heads_q; a.Hkv = (int)d->heads_kv;
    a.Sq = (int)d->seq_q; a.Sk = (int)d->seq_k; a.D = (int)d->head_dim;
    a.scale = resolve_scale(d->scale, d->head_dim);
    a.causal = d->causal != 0;
    a.dtype = d->dtype; a.device = d->device;
    if (const char* e = getenv("AULE_TL")) {
        if (e[0] == 'w' && e[1] == '4')  // one wave per SIMD: 4 waves x 2048 tagged stamps (tools/timeline_w4.py)
            return aule_hip::launch_fwd_w4_timeline(a, stamps, (hipStream_t)d->stream);
    }
    return aule_hip::launch_fwd_pp_timeline(a, stamps, (hipStream_t)d->stream);
}
#endif  // AULE_DEBUG_HOOKS

}  // extern "C"
