/*
 * include/aule.h -- C-ABI of libaule.so (MI355X / gfx950 HIP build).
 *
 * Drop-in boundary for the FlashAttention forward/backward hot path of
 * AuleTechnologies/Aule-Attention.  Every legacy symbol below has the exact
 * name, argument order, scalar widths and return-code convention of the
 * reference export it replaces (reference file:line cited per symbol; the
 * authoritative consumer-side declarations are the ctypes signatures in
 * python/aule/vulkan.py:224-406).  The "_ex" entry points at the bottom are
 * ADDITIVE: they carry what the legacy ABI cannot express (bf16/fp16 storage,
 * GQA/MQA, Sq != Sk, user scale, device pointers, streams).
 *
 * Calling convention: C.  No torch / C++ types cross this boundary.
 * Errors: negative int32 return + text from aule_get_error().
 * Threading: all entry points are serialised by one internal mutex (superset of
 * the reference contract, which has no locking: src/lib.zig:12-22).
 * Legacy compute calls are synchronous (device idle on return), like the
 * reference (src/attention_pipeline.zig:389-390, src/attention_gpu.zig:467-468).
 * "_ex" calls are asynchronous on the caller's HIP stream.
 */
#ifndef AULE_H
#define AULE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AULE_MAX_TENSORS 1024u /* src/lib.zig:17 MAX_TENSORS */

/* ---- lifecycle / info ---------------------------------------------------- */
int32_t aule_init(void);                     /* src/lib.zig:59   0 ok (idempotent), -1 + error text */
void aule_shutdown(void);                    /* src/lib.zig:105  destroys all tensors + context */
const char* aule_get_error(void);            /* src/lib.zig:124  static 512-B buffer, "No error" if none */
const char* aule_get_backend_name(void);     /* src/lib.zig:133  "HIP/ROCm" | "Not initialized" (backend.zig:496-502) */
int32_t aule_get_vendor(void);               /* src/lib.zig:144  1 = amd; -1 uninitialised */
int32_t aule_get_gpu_vendor(void);           /* src/lib.zig:274  same table */
int32_t aule_is_amd_optimized(void);         /* src/lib.zig:168  1 / -1 */
int32_t aule_has_fp16(void);                 /* src/lib.zig:184  1 / -1 */
int32_t aule_get_subgroup_size(void);        /* src/lib.zig:296  64 (wavefront) / -1 */
int32_t aule_get_device_name(uint8_t* buffer, uint32_t buffer_len); /* src/lib.zig:242 bytes copied (NUL-terminated, truncated) / -1 */
int32_t aule_set_shader_variant(uint8_t variant); /* src/lib.zig:202  0, -1 uninit, -2 unavailable (only variant 0 exists here) */
int32_t aule_get_shader_variant(void);            /* src/lib.zig:215 */
int32_t aule_has_shader_variant(uint8_t variant); /* src/lib.zig:226  1/0/-1 */
int32_t aule_supports_backward(void);        /* src/lib.zig:96   1 */

/* ---- host-pointer forward (MHA, Sq == Sk, fp32, scale = 1/sqrt(D)) -------- */
/* src/lib.zig:312-367.  0; -1 uninit; -2 alloc; -3 upload; -4 compute; -5 download */
int32_t aule_attention_forward(const float* query, const float* key, const float* value, float* output,
                               uint32_t batch_size, uint32_t num_heads, uint32_t seq_len,
                               uint32_t head_dim, int32_t causal);

/* ---- persistent device tensors (1-based slot handles, 0 = failure) -------- */
typedef uint64_t aule_tensor_handle;
aule_tensor_handle aule_tensor_create(uint32_t batch_size, uint32_t num_heads, uint32_t seq_len,
                                      uint32_t head_dim);      /* src/lib.zig:409 */
aule_tensor_handle aule_tensor_create_u32(uint32_t batch_size, uint32_t num_heads, uint32_t seq_len,
                                          uint32_t head_dim);  /* src/lib.zig:432 (aliases fp32 storage) */
void aule_tensor_destroy(aule_tensor_handle handle);           /* src/lib.zig:444 ignores 0 / out of range */
int32_t aule_tensor_upload(aule_tensor_handle handle, const float* data, uint32_t count);     /* :457  0; -1 bad handle; -3 size mismatch */
int32_t aule_tensor_download(aule_tensor_handle handle, float* output, uint32_t count);       /* :469 */
int32_t aule_tensor_download_u32(aule_tensor_handle handle, uint32_t* output, uint32_t count);/* :481 */
uint32_t aule_tensor_size(aule_tensor_handle handle);          /* src/lib.zig:626 element count, 0 if invalid */
uint32_t aule_tensor_count(void);                              /* src/lib.zig:383 */
uint32_t aule_tensor_max(void);                                /* src/lib.zig:392 */
void aule_tensor_clear_all(void);                              /* src/lib.zig:397 */

/* ---- handle-based forward: GQA (Hkv from K's shape), cross-attn, causal ---- */
/* src/lib.zig:496-529 -> backend.zig:318-370 -> attention_gpu.zig:360-453.     */
/* rot_cos/rot_sin: both 0 (no rotation) or both handles of fp32 tensors shaped   */
/* [1, 1, >= max(seq_q, seq_k), head_dim/2] (interleaved pairs, one table for all */
/* heads); anything else is -3 + error text.                                      */
/* window_size > 0: sliding window, key j visible to query i only if            */
/* i - j < window_size, on top of the causal rule (the convention of the kernel */
/* the reference runs on ROCm, python/aule/triton_flash_amd.py:179-183).        */
int32_t aule_attention_forward_gpu(aule_tensor_handle q, aule_tensor_handle k, aule_tensor_handle v,
                                   aule_tensor_handle output, aule_tensor_handle rot_cos,
                                   aule_tensor_handle rot_sin, int32_t causal, int32_t window_size);

/* ---- training path (host pointers, MHA, Sq == Sk, fp32) -------------------- */
/* src/lib.zig:765-852: O and LSE[B,H,S] = m + ln(l) of the scaled scores.      */
int32_t aule_attention_forward_with_lse(const float* query, const float* key, const float* value,
                                        float* output, float* lse, uint32_t batch_size,
                                        uint32_t num_heads, uint32_t seq_len, uint32_t head_dim,
                                        int32_t causal);
/* src/lib.zig:639-762: dQ,dK,dV from Q,K,V,O,dO,LSE.                           */
int32_t aule_attention_backward(const float* query, const float* key, const float* value,
                                const float* output, const float* grad_output, const float* lse,
                                float* grad_query, float* grad_key, float* grad_value,
                                uint32_t batch_size, uint32_t num_heads, uint32_t seq_len,
                                uint32_t head_dim, int32_t causal);

/* ---- out-of-scope features: exported so that ctypes attribute lookup in the  */
/* reference binding (vulkan.py:300-316) keeps working; they return -3.        */
int32_t aule_attention_forward_paged(aule_tensor_handle q, aule_tensor_handle k, aule_tensor_handle v,
                                     aule_tensor_handle output, aule_tensor_handle rot_cos,
                                     aule_tensor_handle rot_sin, int32_t causal,
                                     int32_t window_size);                    /* src/lib.zig:533 */
int32_t aule_spatial_sort(aule_tensor_handle keys, aule_tensor_handle values,
                          aule_tensor_handle indices, uint32_t sort_dim);     /* src/lib.zig:568 */
int32_t aule_attention_forward_gravity(aule_tensor_handle q, aule_tensor_handle k, aule_tensor_handle v,
                                       aule_tensor_handle output, aule_tensor_handle rot_cos,
                                       aule_tensor_handle rot_sin, aule_tensor_handle indices,
                                       int32_t causal, uint32_t max_attend,
                                       int32_t window_size);                  /* src/lib.zig:587 */

/* ========================================================================== */
/* Additive entry points (no reference counterpart; they are what the Python   */
/* surface aule.flash_attention binds for torch/ROCm tensors, replacing the    */
/* Triton launch at python/aule/triton_flash_amd.py:393-500).                 */
/* ========================================================================== */
typedef enum aule_dtype {
    AULE_DTYPE_F32 = 0,  /* fp32 storage, fp32 MFMA (v_mfma_f32_32x32x2_f32) */
    AULE_DTYPE_F16 = 1,  /* fp16 storage, fp16 MFMA, fp32 accumulate/softmax */
    AULE_DTYPE_BF16 = 2  /* bf16 storage, bf16 MFMA, fp32 accumulate/softmax */
} aule_dtype;

/* All tensors row-major contiguous: Q,O,dO,dQ [B,Hq,Sq,D]; K,V,dK,dV [B,Hkv,Sk,D];
 * LSE [B,Hq,Sq] fp32.  Pointers are DEVICE pointers on `device`.
 * head_dim in {32, 64, 128}; heads_q % heads_kv == 0.
 * `causal`: 0 = none; 1 = top-left aligned (query i sees keys j <= i), the rule of
 * every reference implementation; 2 = bottom-right aligned (query i sits at position
 * i + seq_k - seq_q, i.e. the last query sees every key; needs seq_k >= seq_q) --
 * an additive option for chunked prefill / decode against a longer KV history.
 * With mode 2 the sliding window is measured from the shifted position as well.
 * The handle-based and host entry points above keep the reference's rule (any
 * non-zero `causal` = top-left). */
#define AULE_CAUSAL_NONE 0
#define AULE_CAUSAL_TOP_LEFT 1
#define AULE_CAUSAL_BOTTOM_RIGHT 2
typedef struct aule_attn_desc {
    uint32_t struct_size;  /* = sizeof(aule_attn_desc) */
    int32_t dtype;         /* aule_dtype */
    uint32_t batch, heads_q, heads_kv, seq_q, seq_k, head_dim;
    float scale;           /* softmax scale; 0 or NaN => 1/sqrt(head_dim) */
    int32_t causal;        /* AULE_CAUSAL_* */
    int32_t window_size;   /* <= 0: full attention; W > 0: key j visible to query i only if i - j < W */
    int32_t device;        /* HIP device ordinal; -1 = current device */
    void* stream;          /* hipStream_t; NULL = default stream */
    const void* q;
    const void* k;
    const void* v;
    void* out;
    float* lse;            /* optional (NULL to skip) */
    void* workspace;       /* optional device buffer, 16-byte aligned, >= aule_attention_forward_workspace_size()  */
    uint64_t workspace_bytes;  /* bytes; NULL / too small: the library allocates stream-ordered (hipMallocAsync), which */
                           /* is correct everywhere but adds costly alloc/free nodes under hipGraph capture         */
} aule_attn_desc;

typedef struct aule_attn_bwd_desc {
    uint32_t struct_size;  /* = sizeof(aule_attn_bwd_desc) */
    int32_t dtype;
    uint32_t batch, heads_q, heads_kv, seq_q, seq_k, head_dim;
    float scale;
    int32_t causal;
    int32_t window_size;
    int32_t device;
    void* stream;
    const void* q;
    const void* k;
    const void* v;
    const void* out;
    const void* dout;
    const float* lse;
    void* dq;
    void* dk;
    void* dv;
    void* workspace;       /* >= aule_attention_backward_workspace_size() bytes, device memory */
    uint64_t workspace_bytes;
} aule_attn_bwd_desc;

/* 0 ok; -1 uninitialised; -3 invalid/unsupported arguments; -4 launch failure. Asynchronous. */
int32_t aule_attention_forward_ex(const aule_attn_desc* desc);
int32_t aule_attention_backward_ex(const aule_attn_bwd_desc* desc);
/* Paged-KV decode (additive).  Replaces python/aule/triton_flash_amd.py:543-737            */
/* (_paged_attention_fwd_amd / flash_attention_paged_amd): one query token per sequence, vLLM-style block tables.  */
/* Asynchronous on `stream`; device pointers.                                                                     */
typedef struct aule_paged_desc {
    uint32_t struct_size;      /* = sizeof(aule_paged_desc) */
    int32_t dtype;             /* AULE_DTYPE_F16 or AULE_DTYPE_BF16 */
    uint32_t batch, heads_q, heads_kv, head_dim;   /* head_dim 32, 64 or 128 */
    uint32_t block_size;       /* tokens per cache block */
    uint32_t max_blocks;       /* columns of block_tables */
    float scale;               /* 0 -> 1/sqrt(head_dim) */
    int32_t window_size;       /* > 0: only the last window_size positions (context_len - 1 - pos < window_size) */
    int32_t device;            /* HIP device ordinal, -1 = current */
    void* stream;              /* hipStream_t */
    const void* q;             /* [batch, heads_q, head_dim] */
    const void* k_cache;       /* [num_blocks, block_size, heads_kv, head_dim] */
    const void* v_cache;       /* same layout */
    const int32_t* block_tables;   /* [batch, max_blocks]: physical block of each logical block */
    const int32_t* context_lens;   /* [batch]: keys per sequence (0 -> output row of zeros); read on the device and clamped
                                      there to [0, max_blocks * block_size], so a stale value cannot index out of the table */
    void* out;                 /* [batch, heads_q, head_dim] */
    void* workspace;           /* optional, as aule_attn_desc.workspace; size from aule_attention_paged_decode_workspace_size() */
    uint64_t workspace_bytes;
} aule_paged_desc;
/* 0 ok; -1 uninitialised; -3 invalid/unsupported arguments; -4 launch failure. */
int32_t aule_attention_paged_decode_ex(const aule_paged_desc* desc);
/* Bytes of optional workspace the forward / paged decode would otherwise allocate itself for this descriptor   */
/* (0: single-launch path, nothing needed).  Host logic only; pointers in the descriptor are not read.           */
uint64_t aule_attention_forward_workspace_size(const aule_attn_desc* desc);
uint64_t aule_attention_paged_decode_workspace_size(const aule_paged_desc* desc);

/* Rotary position embedding pass (additive; SURVEY.md 8f row N1, second half).  Replaces the rotation the           */
/* reference fuses into its kernels: python/aule/triton_flash.py:32-52,:112-131,:165-180 (layout HALF) and            */
/* shaders/attention_f32.comp:98-111,:132-145 (layout INTERLEAVED).  x is [rows_bh, seq, head_dim] with               */
/* `row_pitch` elements per row; row s uses table row s + pos_offset; `inverse` applies the transposed rotation       */
/* (dQ', dK' -> dQ, dK).  in == out is allowed.  Asynchronous on `stream`; device pointers.                           */
#define AULE_ROPE_HALF 0         /* pairs (p, p + head_dim/2) */
#define AULE_ROPE_INTERLEAVED 1  /* pairs (2p, 2p + 1) */
typedef struct aule_rope_desc {
    uint32_t struct_size;      /* = sizeof(aule_rope_desc) */
    int32_t dtype;             /* aule_dtype */
    uint64_t rows_bh;          /* batch * heads */
    uint32_t seq, head_dim;    /* head_dim even (any size; 16-byte vector path when head_dim % 16 == 0) */
    uint32_t row_pitch;        /* elements per row, >= head_dim */
    uint32_t table_len;        /* rows of cos / sin; seq + pos_offset <= table_len */
    uint32_t table_pitch;      /* floats per table row; 0 = head_dim/2 */
    int32_t layout;            /* AULE_ROPE_* */
    int32_t inverse;
    uint32_t pos_offset;
    int32_t device;
    void* stream;
    const void* in;
    void* out;
    const float* cos;          /* [table_len, head_dim/2] fp32 */
    const float* sin;
} aule_rope_desc;
int32_t aule_rope_ex(const aule_rope_desc* desc);

/* Attention with the query rotation fused into the kernel (additive; inference path).  Replaces the Q half of       */
/* python/aule/triton_flash.py:112-131 (the reference rotates Q in its forward prologue); K must arrive rotated --   */
/* once per key by aule_rope_ex(), e.g. when it is appended to a KV cache -- because the tiled kernel re-reads every */
/* K tile once per 256 query rows and would repeat the rotation each time.  Q is read un-rotated and */
/* rotated in registers with exactly aule_rope_ex()'s arithmetic and rounding: the result is bit-identical to        */
/* aule_rope_ex(Q) followed by aule_attention_forward_ex(), minus one read and one write of Q.  Taken only by the    */
/* persistent forward kernel: fp16 / bf16, head_dim 64 or 128, AULE_ROPE_HALF, 16-byte aligned tables with           */
/* table_pitch % 4 == 0, no sliding window other than a causal one of >= 128 keys (round 6: the kernel's window     */
/* instances) -- ask aule_attention_forward_rope_fusable() (host logic only; 1 = yes)                               */
/* and fall back to the two calls otherwise; aule_attention_forward_rope_ex() returns -3 for other configurations.   */
typedef struct aule_attn_rope {
    uint32_t struct_size;      /* = sizeof(aule_attn_rope) */
    int32_t layout;            /* AULE_ROPE_HALF */
    uint32_t table_len;        /* rows of cos / sin; seq_q + q_pos_offset <= table_len */
    uint32_t table_pitch;      /* floats per table row; 0 = head_dim/2 */
    uint32_t q_pos_offset;     /* query i uses table row i + q_pos_offset (seq_k - seq_q for bottom-right causal) */
    uint32_t reserved;
    const float* cos;          /* [table_len, head_dim/2] fp32 */
    const float* sin;
} aule_attn_rope;
int32_t aule_attention_forward_rope_ex(const aule_attn_desc* desc, const aule_attn_rope* rope);
int32_t aule_attention_forward_rope_fusable(const aule_attn_desc* desc, const aule_attn_rope* rope);

uint64_t aule_attention_backward_workspace_size(const aule_attn_bwd_desc* desc);

/* Build/ABI identification: "aule-hip gfx950 <abi>" */
const char* aule_hip_build_info(void);
/* Debug (not part of the drop-in ABI): forward kernel aule_attention_forward_ex would pick for `desc` --        */
/* 0 fp32, 1 ping-pong, 4 split-KV, 5 tiled split, 7 one wave per SIMD over key-range pieces, 8 one wave per SIMD; -3 bad. */
/* Host logic only, no aule_init().                                                                                  */
int32_t aule_hip_debug_forward_route(const aule_attn_desc* desc);
/* Debug: bit mask of the kernels the most recent backward launch of this process ran -- 1 the 5-matmul mode (delta pass, dK/dV kernel  */
/* spilling its dS, dQ = dS K), 2 / 4 the one-wave-per-SIMD dQ / dK/dV kernel, 8 / 16 their two-waves-per-SIMD predecessors, 32 the    */
/* fp32 kernels, 64 (with 4) the D = 64 dK/dV instance with two key blocks per wave; 0 before the first backward.  Lets a test assert  */
/* the mode a shape takes by itself.                                                                                      */
int32_t aule_hip_debug_last_backward_route(void);
/* Debug: the causal-split plan of route 7 (small causal grids) as integers -- out = {pieces n, pairs, then per pair of   */
/* Q blocks: tiles of the far block, of the near block, cut positions b[0..8]}; returns the ints written (negative: the */
/* capacity needed), 0 when the shape does not take that route.  Host logic only.                                       */
int32_t aule_hip_debug_forward_split_plan(const aule_attn_desc* desc, int32_t* out, int32_t cap);
/* Debug: the work item of workgroup `bid` of a grid of batch * heads_q * nblk workgroups -- out4 = {batch, kv head, q head,  */
/* block}.  ranked = 0: unit by unit (flag: last block first), the 16-bit kernels; ranked = 1: block rank by block rank over  */
/* all units (flag: descending), the fp32 kernels.  Host logic only; -3 on a bad argument.                                    */
int32_t aule_hip_debug_work_order(int32_t ranked, int32_t bid, int32_t batch, int32_t heads_q, int32_t heads_kv, int32_t nblk, int32_t flag, int32_t* out4);

/* ---- Direct peer exchange between the per-GPU processes of one node (additive; no counterpart in the reference, which  */
/* is single-device: SURVEY.md 8e).  A rank allocates its receive buffer here, publishes the 64-byte handle by any means    */
/* (aule.dist uses the process group), opens its peers' handles, and copies pieces of its output straight into their        */
/* buffers over xGMI -- no RCCL algorithm choice in the data path.  Every call: 0, or -1 bad argument / -2 allocation /     */
/* -4 HIP error with the text in aule_get_error().  No aule_init() needed; `device` < 0 = the current device.               */
typedef struct aule_ipc_handle { unsigned char bytes[64]; } aule_ipc_handle;   /* = hipIpcMemHandle_t */
int32_t aule_peer_alloc(int32_t device, uint64_t bytes, void** ptr, aule_ipc_handle* handle);   /* hipMalloc + hipIpcGetMemHandle */
int32_t aule_peer_free(int32_t device, void* ptr);
int32_t aule_peer_open(int32_t device, const aule_ipc_handle* handle, void** ptr);              /* another process's buffer, mapped here */
int32_t aule_peer_close(int32_t device, void* ptr);
/* dst / src: device pointers (local, or a peer's from aule_peer_open) + byte offsets applied by the caller; asynchronous on */
/* `stream` (a hipStream_t of `device`).                                                                                      */
int32_t aule_peer_copy_async(int32_t device, void* dst, const void* src, uint64_t bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* AULE_H */
