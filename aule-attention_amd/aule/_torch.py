"""PyTorch-ROCm binding of the gfx950 kernels (zero-copy, current stream, autograd).

Counterpart of the reference's FlashAttentionAMDFunc (python/aule/triton_flash_amd.py:
392-500) and flash_attention_amd (:503-536): same autograd contract -- forward saves
(q, k, v, out, lse), backward returns (dq, dk, dv) in the input dtype -- but the launches
go to libaule.so's aule_attention_forward_ex / aule_attention_backward_ex with
tensor.data_ptr() and the current HIP stream.  torch is plumbing here (device memory,
streams, autograd graph); the arithmetic is in aule-attention_amd/csrc/*.hip.
"""
import ctypes
import math
import os
import threading

import torch

from . import _capi

_DTYPES = {torch.float32: _capi.DTYPE_F32, torch.float16: _capi.DTYPE_F16, torch.bfloat16: _capi.DTYPE_BF16}
SUPPORTED_HEAD_DIMS = (32, 64, 128)


_ALWAYS_AUTOGRAD = os.environ.get("AULE_HIP_ALWAYS_AUTOGRAD", "0") == "1"


def _stream_ptr(device):
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def causal_code(causal):
    """The C-ABI's AULE_CAUSAL_* code for the Python-level `causal` argument: False/None -> 0,
    True or "top-left" -> 1 (the reference's rule), "bottom-right" or 2 -> 2."""
    if isinstance(causal, str):
        try:
            return {"none": 0, "top-left": 1, "bottom-right": 2}[causal]
        except KeyError:
            raise ValueError(f"causal must be a bool, 'top-left' or 'bottom-right', got {causal!r}") from None
    if causal is None or causal is False:
        return 0
    if causal is True:
        return 1
    c = int(causal)
    if c not in (0, 1, 2):
        raise ValueError(f"causal code must be 0, 1 or 2, got {causal!r}")
    return c


def _abi_scale(scale):
    """The `_ex` descriptors read scale == 0 as "default 1/sqrt(D)" (a zero-initialised struct must work); this layer always
    resolves the default itself, so an explicit 0.0 -- uniform attention in the reference and in the oracle -- is passed
    as the smallest value the kernels treat the same way (they clamp |scale * log2 e| to 1e-30 anyway)."""
    scale = float(scale)
    return scale if scale != 0.0 else 1e-30


def _same_device(what, ref, *tensors):
    for t in tensors:
        if t.device != ref.device:
            raise ValueError(f"{what}: every tensor must live on {ref.device}, got one on {t.device}")


def _workspace(nbytes, device):
    """Partials of the two-launch short-query paths from torch's caching allocator: stream-ordered like the library's own
    hipMallocAsync fallback, but graph-aware -- under torch.cuda.graph capture it adds no alloc / free nodes (which cost
    more than the kernels of a decode step).  Freed by reference count after the launch; the allocator keeps the block
    on the launch stream's free list, so reuse is ordered behind the kernels that read it."""
    nbytes = int(nbytes)
    return torch.empty((nbytes,), device=device, dtype=torch.uint8) if nbytes > 0 else None


def _attn_rope(cos, sin, q_pos):
    r = _capi.AttnRope()
    r.struct_size = ctypes.sizeof(_capi.AttnRope)
    r.layout = _capi.ROPE_HALF
    r.table_len, r.table_pitch, r.q_pos_offset = cos.shape[0], cos.stride(0), int(q_pos)
    r.cos, r.sin = cos.data_ptr(), sin.data_ptr()
    return r


def fwd_raw(q, k, v, causal, scale, want_lse=True, window=-1, q_rope=None, out=None):
    """q [B,Hq,Sq,D], k/v [B,Hkv,Sk,D]: contiguous device tensors, D in SUPPORTED_HEAD_DIMS.
    Returns (out, lse or None).  Asynchronous on the current stream.
    out: write the result there (contiguous, q's shape / dtype / device) instead of a fresh tensor -- aule.dist computes its
    pieces straight into their place in the gathered tensor.
    q_rope = (cos, sin, q_pos): rotate Q inside the kernel (half-split pairs; K already rotated) -- only for shapes
    rope_fusable() accepts, AuleError otherwise."""
    lib = _capi.get_lib()
    _same_device("flash attention forward", q, k, v)
    B, Hq, Sq, D = q.shape
    if out is None:
        out = torch.empty_like(q)
    elif out.shape != q.shape or out.dtype != q.dtype or out.device != q.device or not out.is_contiguous():
        raise ValueError("out must be a contiguous tensor of the query's shape, dtype and device")
    lse = torch.empty((B, Hq, Sq), device=q.device, dtype=torch.float32) if want_lse else None
    if q.numel() == 0:
        return out, lse
    d, ws_bytes = _fwd_desc(lib, q, k, causal, scale, window)
    d.stream = torch.cuda.current_stream(q.device).cuda_stream
    d.q, d.k, d.v, d.out = q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr()
    d.lse = lse.data_ptr() if lse is not None else None
    if ws_bytes:
        ws = torch.empty((ws_bytes,), device=q.device, dtype=torch.uint8)
        d.workspace, d.workspace_bytes = ws.data_ptr(), ws_bytes
    if q_rope is not None:
        r = _attn_rope(*q_rope)
        _capi.check(lib.aule_attention_forward_rope_ex(ctypes.byref(d), ctypes.byref(r)), "aule_attention_forward_rope_ex")
    else:
        rc = lib.aule_attention_forward_ex(ctypes.byref(d))
        if rc != 0:
            _capi.check(rc, "aule_attention_forward_ex")
    return out, lse


# Per-call host cost (tools/host_overhead.py): the descriptor of a (dtype, shape, flags, device) combination is built
# once -- struct fill, flag decoding, the workspace-size query -- and kept; a call then sets the stream, four or five pointers and the
# workspace.  Per thread (a descriptor is mutated by the call that uses it).  Decode loops that need less than this layer can give
# (~15 us of Python + ctypes + two tensor allocations per call) capture the step in a hipGraph: INTEGRATION.md, tests/test_gpu_graph.py.
_desc_cache = threading.local()


def _fwd_desc(lib, q, k, causal, scale, window):
    cache = getattr(_desc_cache, "fwd", None)
    if cache is None:
        cache = _desc_cache.fwd = {}
    # (the key holds VALUES -- a 0-dim tensor or numpy scalar as `scale` / `window` hashes by identity, never hits and stays alive in
    # the cache -- and the RESOLVED device: an index-less "cuda" device means the current one at the time of the call)
    dev = q.device.index if q.device.index is not None else torch.cuda.current_device()
    key = (q.dtype, tuple(q.shape), int(k.shape[1]), int(k.shape[2]), causal_code(causal), None if scale is None else float(scale),
           int(window) if window is not None and window > 0 else -1, dev)
    ent = cache.get(key)
    if ent is None:
        B, Hq, Sq, D = q.shape
        d = _capi.AttnDesc()
        d.struct_size = ctypes.sizeof(_capi.AttnDesc)
        d.dtype = _DTYPES[q.dtype]
        d.batch, d.heads_q, d.heads_kv, d.seq_q, d.seq_k, d.head_dim = B, Hq, k.shape[1], Sq, k.shape[2], D
        d.scale = _abi_scale(scale)
        d.causal = causal_code(causal)
        d.window_size = int(window) if window is not None and window > 0 else -1
        d.device = dev
        if len(cache) >= 512:
            cache.clear()
        ent = cache[key] = (d, int(lib.aule_attention_forward_workspace_size(ctypes.byref(d))))
    return ent


def rope_fusable(q, k, causal, window, cos, sin, q_pos, scale=None):
    """Would the forward kernel rotate Q itself for this problem (aule_attention_forward_rope_fusable)?  Host logic only:
    no device, no aule_init().  scale: the softmax scale of the call (None = default); it is part of the answer -- the kernel that
    rotates Q does not take negative scales (round 5: without it, flash_attention_rope(scale < 0) chose the fused form in inference
    and the launch then refused it; found by tools/fuzz_parity.py split)."""
    lib = _capi.load()
    if q.dtype not in _DTYPES or q.numel() == 0 or cos.stride(-1) != 1:
        return False
    d = _capi.AttnDesc()
    d.struct_size = ctypes.sizeof(_capi.AttnDesc)
    d.dtype = _DTYPES[q.dtype]
    d.batch, d.heads_q, d.heads_kv, d.seq_q, d.seq_k, d.head_dim = q.shape[0], q.shape[1], k.shape[1], q.shape[2], k.shape[2], q.shape[3]
    d.causal = causal_code(causal)
    d.window_size = int(window) if window is not None and window > 0 else -1
    d.scale = 0.0 if scale is None else _abi_scale(scale)
    r = _attn_rope(cos, sin, q_pos)
    return lib.aule_attention_forward_rope_fusable(ctypes.byref(d), ctypes.byref(r)) == 1


def bwd_raw(q, k, v, out, dout, lse, causal, scale, window=-1):
    lib = _capi.get_lib()
    _same_device("flash attention backward", q, k, v, out, dout, lse)
    B, Hq, Sq, D = q.shape
    Hkv, Sk = k.shape[1], k.shape[2]
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    if q.numel() == 0 or k.numel() == 0:
        return dq.zero_(), dk.zero_(), dv.zero_()
    d = _capi.AttnBwdDesc()
    d.struct_size = ctypes.sizeof(_capi.AttnBwdDesc)
    d.dtype = _DTYPES[q.dtype]
    d.batch, d.heads_q, d.heads_kv, d.seq_q, d.seq_k, d.head_dim = B, Hq, Hkv, Sq, Sk, D
    d.scale = _abi_scale(scale)
    d.causal = causal_code(causal)
    d.window_size = int(window) if window is not None and window > 0 else -1
    d.device = q.device.index if q.device.index is not None else torch.cuda.current_device()
    d.stream = _stream_ptr(q.device)
    d.q, d.k, d.v, d.out, d.dout, d.lse = (q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(),
                                            dout.data_ptr(), lse.data_ptr())
    d.dq, d.dk, d.dv = dq.data_ptr(), dk.data_ptr(), dv.data_ptr()
    # delta [B,Hq,Sq] fp32 (+ fp32 dK/dV partials when the GQA head loop is split over workgroups)
    nbytes = int(lib.aule_attention_backward_workspace_size(ctypes.byref(d)))
    ws = torch.empty((nbytes,), device=q.device, dtype=torch.uint8)
    d.workspace, d.workspace_bytes = ws.data_ptr(), nbytes
    _capi.check(lib.aule_attention_backward_ex(ctypes.byref(d)), "aule_attention_backward_ex")
    return dq, dk, dv


class FlashAttentionHipFunc(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, k, v, causal, scale, window=-1):
        q, k, v = q.contiguous(), k.contiguous(), v.contiguous()
        out, lse = fwd_raw(q, k, v, causal, scale, want_lse=True, window=window)
        ctx.save_for_backward(q, k, v, out, lse)      # triton_flash_amd.py:436
        ctx.causal, ctx.scale, ctx.window = causal, scale, window
        return out

    @staticmethod
    def backward(ctx, dout):
        q, k, v, out, lse = ctx.saved_tensors
        dout = dout.contiguous().to(q.dtype)
        # (the reference's backward drops the window, triton_flash_amd.py:447-500; here it is the true gradient)
        dq, dk, dv = bwd_raw(q, k, v, out, dout, lse, ctx.causal, ctx.scale, window=ctx.window)
        return dq, dk, dv, None, None, None


def _pad_head_dim(x, Dp):
    D = x.shape[-1]
    return x if D == Dp else torch.nn.functional.pad(x, (0, Dp - D))


def flash_attention_hip(q, k, v, causal=True, scale=None, window=-1):
    """Device tensors in, device tensor out, autograd-aware.  dtype fp16/bf16/fp32 run
    natively; anything else is computed in fp32 and cast back."""
    D = q.shape[-1]
    if D > 128:
        raise ValueError(f"head_dim must be <= 128 for the HIP backend, got {D}")
    if scale is None:
        scale = 1.0 / math.sqrt(D)
    orig_dtype = q.dtype
    if orig_dtype not in _DTYPES:
        q, k, v = q.float(), k.float(), v.float()
    if k.dtype != q.dtype or v.dtype != q.dtype:
        k, v = k.to(q.dtype), v.to(q.dtype)
    Dp = next(x for x in SUPPORTED_HEAD_DIMS if x >= D)
    if Dp != D:   # zero-padded head dim: dot products and outputs are unchanged
        q, k, v = _pad_head_dim(q, Dp), _pad_head_dim(k, Dp), _pad_head_dim(v, Dp)
    code = causal_code(causal)
    if code == 2 and k.shape[2] < q.shape[2]:
        raise ValueError(f"bottom-right causal alignment needs seq_len_k >= seq_len_q, got {k.shape[2]} < {q.shape[2]}")
    if torch.is_grad_enabled() and (q.requires_grad or k.requires_grad or v.requires_grad):
        out = FlashAttentionHipFunc.apply(q, k, v, code, float(scale), int(window))
    else:
        # inference: no autograd node (its apply() costs more CPU time than a decode kernel runs) and no LSE,
        # which only the backward reads.  AULE_HIP_ALWAYS_AUTOGRAD=1 restores the old route (A/B measurements).
        if _ALWAYS_AUTOGRAD:
            out = FlashAttentionHipFunc.apply(q, k, v, code, float(scale), int(window))
        else:
            out, _ = fwd_raw(q.contiguous(), k.contiguous(), v.contiguous(), code, float(scale), want_lse=False,
                             window=int(window))
    if Dp != D:
        out = out[..., :D]
    return out if out.dtype == orig_dtype else out.to(orig_dtype)


_ROPE_LAYOUTS = {"half": _capi.ROPE_HALF, "interleaved": _capi.ROPE_INTERLEAVED}


def _rope_tables(cos, sin, D, device):
    """cos / sin as contiguous fp32 [table_len, D/2] on `device` (the reference takes [S, D/2] or [1, S, D/2]:
    triton_flash.py:414-424)."""
    if cos is None or sin is None:
        raise ValueError("cos and sin are required for RoPE")
    if cos.shape[-1] != D // 2 or sin.shape[-1] != D // 2:
        raise ValueError(f"cos/sin must have shape [..., {D // 2}], got {tuple(cos.shape)} / {tuple(sin.shape)}")
    cos = cos.to(device=device, dtype=torch.float32).reshape(-1, D // 2).contiguous()
    sin = sin.to(device=device, dtype=torch.float32).reshape(-1, D // 2).contiguous()
    if cos.shape != sin.shape:
        raise ValueError("cos and sin must have the same shape")
    return cos, sin


def rope_raw(x, cos, sin, layout="half", inverse=False, pos_offset=0, out=None):
    """Rotary embedding pass on the HIP backend (csrc/rope_gfx950.hip): x [B,H,S,D] contiguous device tensor
    (fp32/fp16/bf16), cos/sin fp32 [table_len, D/2]; row s uses table row s + pos_offset.  `out` may be x."""
    lib = _capi.get_lib()
    B, H, S, D = x.shape
    if D % 2:
        raise ValueError(f"RoPE needs an even head_dim, got {D}")
    if out is None:
        out = torch.empty_like(x)
    if x.numel() == 0:
        return out
    if S + pos_offset > cos.shape[0]:
        raise ValueError(f"RoPE table has {cos.shape[0]} rows, sequence needs {S + pos_offset}")
    d = _capi.RopeDesc()
    d.struct_size = ctypes.sizeof(_capi.RopeDesc)
    d.dtype = _DTYPES[x.dtype]
    d.rows_bh, d.seq, d.head_dim, d.row_pitch = B * H, S, D, D
    d.table_len, d.table_pitch = cos.shape[0], 0
    d.layout, d.inverse, d.pos_offset = _ROPE_LAYOUTS[layout], 1 if inverse else 0, int(pos_offset)
    d.device = x.device.index if x.device.index is not None else torch.cuda.current_device()
    d.stream = _stream_ptr(x.device)
    d.in_, d.out, d.cos, d.sin = x.data_ptr(), out.data_ptr(), cos.data_ptr(), sin.data_ptr()
    _capi.check(lib.aule_rope_ex(ctypes.byref(d)), "aule_rope_ex")
    return out


class FlashAttentionRopeHipFunc(torch.autograd.Function):
    """RoPE pass on Q and K, then the attention kernels on the rotated tensors.  The backward rotates dQ', dK' back
    (the rotation is orthogonal: its transpose is the rotation by the opposite angle) -- the reference's backward
    ignores the rotation altogether (triton_flash.py:479-526 differentiates the un-rotated q, k)."""

    @staticmethod
    def forward(ctx, q, k, v, cos, sin, causal, scale, window, layout, q_pos, Dp):
        qr = rope_raw(q.contiguous(), cos, sin, layout, False, q_pos)
        kr = rope_raw(k.contiguous(), cos, sin, layout, False, 0)
        D = q.shape[-1]
        if Dp != D:
            qr, kr, v = _pad_head_dim(qr, Dp), _pad_head_dim(kr, Dp), _pad_head_dim(v, Dp)
        v = v.contiguous()
        out, lse = fwd_raw(qr, kr, v, causal, scale, want_lse=True, window=window)
        ctx.save_for_backward(qr, kr, v, out, lse, cos, sin)
        ctx.args = (causal, scale, window, layout, q_pos, D)
        return out[..., :D] if Dp != D else out

    @staticmethod
    def backward(ctx, dout):
        qr, kr, v, out, lse, cos, sin = ctx.saved_tensors
        causal, scale, window, layout, q_pos, D = ctx.args
        Dp = qr.shape[-1]
        dout = dout.to(qr.dtype)
        dout = _pad_head_dim(dout, Dp).contiguous() if Dp != D else dout.contiguous()
        dq, dk, dv = bwd_raw(qr, kr, v, out, dout, lse, causal, scale, window=window)
        if Dp != D:
            dq, dk, dv = dq[..., :D].contiguous(), dk[..., :D].contiguous(), dv[..., :D]
        rope_raw(dq, cos, sin, layout, True, q_pos, out=dq)
        rope_raw(dk, cos, sin, layout, True, 0, out=dk)
        return dq, dk, dv, None, None, None, None, None, None, None, None


def flash_attention_rope_hip(q, k, v, cos, sin, causal=True, scale=None, window=-1, layout="half"):
    """RoPE + attention on device tensors, autograd-aware.  Query i uses table row i (row i + Sk - Sq with
    causal="bottom-right"), key j row j -- the positions of the reference's kernel (triton_flash.py:119, :169)."""
    D = q.shape[-1]
    if D > 128:
        raise ValueError(f"head_dim must be <= 128 for the HIP backend, got {D}")
    if D % 2:
        raise ValueError(f"RoPE needs an even head_dim, got {D}")
    if layout not in _ROPE_LAYOUTS:
        raise ValueError(f"layout must be 'half' or 'interleaved', got {layout!r}")
    if scale is None:
        scale = 1.0 / math.sqrt(D)
    orig_dtype = q.dtype
    if orig_dtype not in _DTYPES:
        q, k, v = q.float(), k.float(), v.float()
    if k.dtype != q.dtype or v.dtype != q.dtype:
        k, v = k.to(q.dtype), v.to(q.dtype)
    cos, sin = _rope_tables(cos, sin, D, q.device)
    code = causal_code(causal)
    Sq, Sk = q.shape[2], k.shape[2]
    if code == 2 and Sk < Sq:
        raise ValueError(f"bottom-right causal alignment needs seq_len_k >= seq_len_q, got {Sk} < {Sq}")
    q_pos = Sk - Sq if code == 2 else 0
    if max(Sq + q_pos, Sk) > cos.shape[0]:
        raise ValueError(f"RoPE table has {cos.shape[0]} rows, sequences need {max(Sq + q_pos, Sk)}")
    Dp = next(x for x in SUPPORTED_HEAD_DIMS if x >= D)
    needs_grad = torch.is_grad_enabled() and (q.requires_grad or k.requires_grad or v.requires_grad)
    if (not needs_grad and layout == "half" and Dp == D and os.environ.get("AULE_HIP_ROPE_FUSE", "1") != "0"
            and rope_fusable(q, k, code, window, cos, sin, q_pos, float(scale))):
        # inference: K is rotated once per key, Q on its way into the attention kernel's registers (one read and one
        # write of Q less; bit-identical to the two-pass form -- DESIGN.md 3.6).  The backward needs the rotated Q in
        # memory, so training keeps the separate pass.
        kr = rope_raw(k.contiguous(), cos, sin, layout, False, 0)
        out, _ = fwd_raw(q.contiguous(), kr, v.contiguous(), code, float(scale), want_lse=False, window=window,
                         q_rope=(cos, sin, q_pos))
        return out if out.dtype == orig_dtype else out.to(orig_dtype)
    out = FlashAttentionRopeHipFunc.apply(q, k, v, cos, sin, code, float(scale), int(window), layout, q_pos, Dp)
    return out if out.dtype == orig_dtype else out.to(orig_dtype)


def paged_decode(q, k_cache, v_cache, block_tables, context_lens, scale=None, window_size=-1):
    """Paged-KV decode on the HIP backend; counterpart of flash_attention_paged_amd
    (python/aule/triton_flash_amd.py:656-737), same argument meaning:

        q            [batch, heads_q, head_dim] (or [batch, heads_q, 1, head_dim]) fp16 / bf16
        k_cache      [num_blocks, block_size, heads_kv, head_dim]     v_cache: same
        block_tables [batch, max_blocks_per_seq] integer, context_lens [batch] integer
    Returns [batch, heads_q, head_dim].  No device->host synchronisation (the reference reads
    context_lens.max() on the host)."""
    if q.dim() == 4:
        if q.shape[2] != 1:
            raise ValueError("PagedAttention only supports single query token")
        q = q.squeeze(2)
    if q.dim() != 3 or k_cache.dim() != 4 or v_cache.shape != k_cache.shape:
        raise ValueError("expected q [B,Hq,D] and k_cache/v_cache [num_blocks, block_size, Hkv, D]")
    B, Hq, D = q.shape
    _, block_size, Hkv, Dk = k_cache.shape
    if Dk != D:
        raise ValueError(f"head_dim mismatch: query={D}, cache={Dk}")
    if Hq % Hkv != 0:
        raise ValueError(f"heads_q ({Hq}) must be divisible by heads_kv ({Hkv})")
    if q.dtype not in (torch.float16, torch.bfloat16) or k_cache.dtype != q.dtype or v_cache.dtype != q.dtype:
        raise ValueError("paged decode runs in fp16 or bf16 (query and caches in the same dtype)")
    if D not in SUPPORTED_HEAD_DIMS:
        raise ValueError(f"head_dim must be one of {SUPPORTED_HEAD_DIMS} for paged decode, got {D}")
    if scale is None:
        scale = 1.0 / math.sqrt(D)
    lib = _capi.get_lib()
    q, k_cache, v_cache = q.contiguous(), k_cache.contiguous(), v_cache.contiguous()
    _same_device("paged decode", q, k_cache, v_cache)
    # block tables / lengths are read by the kernel: a CPU (or other-GPU) tensor would hand it a foreign pointer
    bt = block_tables.to(device=q.device, dtype=torch.int32).contiguous()
    cl = context_lens.to(device=q.device, dtype=torch.int32).contiguous()
    if bt.dim() != 2 or bt.shape[0] != B or cl.shape != (B,):
        raise ValueError("block_tables must be [batch, max_blocks] and context_lens [batch]")
    out = torch.empty_like(q)
    if B * Hq == 0:
        return out
    d = _capi.PagedDesc()
    d.struct_size = ctypes.sizeof(_capi.PagedDesc)
    d.dtype = _DTYPES[q.dtype]
    d.batch, d.heads_q, d.heads_kv, d.head_dim = B, Hq, Hkv, D
    d.block_size, d.max_blocks = block_size, bt.shape[1]
    d.scale = _abi_scale(scale)
    d.window_size = int(window_size) if window_size is not None and window_size > 0 else -1
    d.device = q.device.index if q.device.index is not None else torch.cuda.current_device()
    d.stream = _stream_ptr(q.device)
    d.q, d.k_cache, d.v_cache, d.out = q.data_ptr(), k_cache.data_ptr(), v_cache.data_ptr(), out.data_ptr()
    ws = _workspace(lib.aule_attention_paged_decode_workspace_size(ctypes.byref(d)), q.device)
    if ws is not None:
        d.workspace, d.workspace_bytes = ws.data_ptr(), ws.numel()
    d.block_tables, d.context_lens = bt.data_ptr(), cl.data_ptr()
    _capi.check(lib.aule_attention_paged_decode_ex(ctypes.byref(d)), "aule_attention_paged_decode_ex")
    return out
