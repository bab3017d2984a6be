"""aule (MI355X / gfx950 HIP build) -- drop-in for the hot path of aule-attention.

Keeps the public surface of the reference's python/aule/__init__.py for the
FlashAttention forward/backward path:

    flash_attention(query, key, value, rot_cos=None, rot_sin=None, causal=True,
                    scale=None, window_size=-1)                    (__init__.py:104)

with the same positional order, defaults, validation (ValueError conditions of
__init__.py:140-160) and container behaviour (torch in -> torch out on the same
device/dtype and inside autograd; NumPy in -> NumPy out).  There is ONE backend:
hand-written HIP kernels for gfx950 behind libaule.so.  No Triton, no Vulkan, no CPU
fallback -- when the library or a HIP device is missing the call raises AuleError.

Also carried: the SDPA shim install() / uninstall() / scaled_dot_product_attention
(__init__.py:288-442).  Not carried over (out of scope): ComfyUI
glue, gravity/sort features.  RoPE: flash_attention_rope / precompute_rope_frequencies / apply_rope_separate
(triton_flash.py:561-703) run a rotation pass + the attention kernels; flash_attention() itself keeps the
behaviour of the reference's ROCm route and ignores rot_cos / rot_sin with a warning.  Sliding window (window_size > 0) follows the convention of the
kernel the reference runs on ROCm (triton_flash_amd.py:179-183): key j is visible to query i only if
i - j < window_size, on top of the causal rule; unlike the reference, the backward honours it too.
"""
import logging
import math
import warnings

from ._capi import AuleError

__version__ = "0.5.0+hip.gfx950"
logger = logging.getLogger(__name__)

_verbose = False


def _validate(query, key, value):
    """Shape rules of the reference, same messages (__init__.py:140-160)."""
    if query.ndim != 4:
        raise ValueError(f"query must be 4D [batch, heads, seq_len, head_dim], got shape {query.shape}")
    if key.ndim != 4:
        raise ValueError(f"key must be 4D [batch, heads, seq_len, head_dim], got shape {key.shape}")
    if value.ndim != 4:
        raise ValueError(f"value must be 4D [batch, heads, seq_len, head_dim], got shape {value.shape}")
    batch_q, heads_q, seq_q, head_dim_q = query.shape
    batch_k, heads_kv, seq_k, head_dim_k = key.shape
    batch_v, heads_v, seq_v, head_dim_v = value.shape
    if batch_q != batch_k or batch_q != batch_v:
        raise ValueError(f"Batch size mismatch: query={batch_q}, key={batch_k}, value={batch_v}")
    if head_dim_q != head_dim_k or head_dim_q != head_dim_v:
        raise ValueError(f"head_dim mismatch: query={head_dim_q}, key={head_dim_k}, value={head_dim_v}")
    if seq_k != seq_v:
        raise ValueError(f"Key/value seq_len mismatch: key={seq_k}, value={seq_v}")
    if heads_kv != heads_v:
        raise ValueError(f"Key/value heads mismatch: key={heads_kv}, value={heads_v}")
    if heads_q % heads_kv != 0:
        raise ValueError(f"heads_q ({heads_q}) must be divisible by heads_kv ({heads_kv}) for GQA")


def _hip_device():
    import torch
    if not torch.cuda.is_available():
        raise AuleError("aule (HIP build): no ROCm device visible to PyTorch; there is no CPU fallback")
    return torch.device("cuda", torch.cuda.current_device())


def flash_attention(query, key, value, rot_cos=None, rot_sin=None, causal=True, scale=None, window_size=-1):
    """FlashAttention-2 on MI355X.

    Args:
        query: [batch, heads_q, seq_len_q, head_dim] torch.Tensor or numpy.ndarray
        key, value: [batch, heads_kv, seq_len_k, head_dim]
        rot_cos, rot_sin: accepted for signature compatibility; ignored with a warning
            (the reference's ROCm route drops them too: __init__.py:204).
        causal: True = top-left aligned causal mask (query i sees keys j <= i), the reference's rule;
            "bottom-right" = query i sits at position i + seq_len_k - seq_len_q (the last query sees every
            key; needs seq_len_k >= seq_len_q) -- an additive option, not in the reference
        scale: softmax scale, default 1/sqrt(head_dim)
        window_size: -1 = full attention; W > 0 = sliding window, key j visible to query i only if i - j < W

    Returns: tensor/array shaped like `query`, same container type and dtype.
    Raises: ValueError for invalid shapes; AuleError if the HIP backend is unavailable.
    """
    import numpy as np
    try:
        import torch
        is_torch = isinstance(query, torch.Tensor)
    except ImportError as e:  # torch is the device-memory plumbing of this build
        raise AuleError("aule (HIP build) needs PyTorch-ROCm for device memory") from e

    _validate(query, key, value)

    if rot_cos is not None or rot_sin is not None:
        warnings.warn("RoPE is not fused in the HIP backend, ignoring rot_cos/rot_sin", stacklevel=2)
    window = int(window_size) if window_size is not None and window_size > 0 else -1

    from ._torch import flash_attention_hip

    if _verbose:
        print(f"aule-attention: hip | shape={tuple(query.shape)} | causal={causal}")

    if is_torch:
        if query.is_cuda:
            if not (key.is_cuda and value.is_cuda):
                raise ValueError("query, key and value must be on the same device")
            return flash_attention_hip(query, key, value, causal=causal, scale=scale, window=window)
        # CPU torch tensor: the reference round-trips through its device backend and
        # returns a tensor on query.device (__init__.py:210-229); same here.
        dev = _hip_device()
        with torch.no_grad():
            out = flash_attention_hip(query.to(dev), key.to(dev), value.to(dev), causal=causal, scale=scale, window=window)
        return out.to(query.device)

    # NumPy in -> NumPy out (dtype follows the input, like _cpu_attention: __init__.py:247-271)
    dev = _hip_device()
    in_dtype = query.dtype
    comp = np.float16 if in_dtype == np.float16 else np.float32
    tq = torch.from_numpy(np.ascontiguousarray(query, dtype=comp)).to(dev)
    tk = torch.from_numpy(np.ascontiguousarray(key, dtype=comp)).to(dev)
    tv = torch.from_numpy(np.ascontiguousarray(value, dtype=comp)).to(dev)
    with torch.no_grad():
        out = flash_attention_hip(tq, tk, tv, causal=causal, scale=scale, window=window)
    out_np = out.cpu().numpy()
    return out_np if out_np.dtype == in_dtype else out_np.astype(in_dtype)


# Alias for compatibility (__init__.py:275)
attention = flash_attention


def flash_attention_paged_amd(q, k_cache, v_cache, block_tables, context_lens, scale=None, window_size=-1):
    """PagedAttention for the decode phase (one query token per sequence, vLLM-style block tables); same name,
    arguments and result as the reference's export (python/aule/triton_flash_amd.py:656-737, __init__.py:59):

        q [batch, heads_q, head_dim]; k_cache, v_cache [num_blocks, block_size, heads_kv, head_dim];
        block_tables [batch, max_blocks_per_seq]; context_lens [batch]  ->  [batch, heads_q, head_dim]

    window_size > 0 keeps only the last window_size positions of each sequence.  fp16 / bf16 ROCm tensors."""
    try:
        import torch  # noqa: F401
    except ImportError as e:
        raise AuleError("aule (HIP build) needs PyTorch-ROCm for device memory") from e
    if not q.is_cuda:
        raise AuleError("aule (HIP build): paged decode needs ROCm device tensors; there is no CPU fallback")
    from ._torch import paged_decode
    return paged_decode(q, k_cache, v_cache, block_tables, context_lens, scale=scale, window_size=window_size)


flash_attention_paged = flash_attention_paged_amd


# =============================================================================
# RoPE (reference python/aule/triton_flash.py:561-703, exported at __init__.py:72-75)
# =============================================================================
def flash_attention_rope(q, k, v, cos, sin, causal=True, scale=None, window_size=-1):
    """RoPE + FlashAttention-2; same name, arguments and result as the reference's export
    (triton_flash.py:561-603): half-split pairs, x_rot = x * cos + rotate_half(x) * sin, query i at table row i,
    key j at row j; cos / sin [seq_len, head_dim // 2] or [1, seq_len, head_dim // 2].

    On MI355X K is rotated by one HBM-streaming pass (csrc/rope_gfx950.hip) -- once per key, not once per Q block.  Q is
    rotated by the same pass when gradients are needed (the backward kernels read the rotated Q, and return the true
    gradients, rotated back -- the reference's backward does not) and inside the forward kernel, on its way into the
    registers, otherwise (bit-identical, one read and one write of Q less: DESIGN.md 3.6).  ROCm tensors; autograd-aware."""
    try:
        import torch  # noqa: F401
    except ImportError as e:
        raise AuleError("aule (HIP build) needs PyTorch-ROCm for device memory") from e
    _validate(q, k, v)
    if cos is None or sin is None:
        raise ValueError("cos and sin are required for RoPE")
    if not (q.is_cuda and k.is_cuda and v.is_cuda):
        raise AuleError("aule (HIP build): flash_attention_rope needs ROCm device tensors; there is no CPU fallback")
    window = int(window_size) if window_size is not None and window_size > 0 else -1
    from ._torch import flash_attention_rope_hip
    return flash_attention_rope_hip(q, k, v, cos, sin, causal=causal, scale=scale, window=window, layout="half")


def precompute_rope_frequencies(seq_len, head_dim, base=10000.0, device="cuda", dtype=None):
    """cos, sin [seq_len, head_dim // 2]: theta_p = base^(-p / (head_dim/2)), angle = position * theta_p
    (same signature and values as triton_flash.py:644-677)."""
    import torch
    dtype = torch.float32 if dtype is None else dtype
    half_dim = head_dim // 2
    freqs = 1.0 / (base ** (torch.arange(0, half_dim, device=device, dtype=dtype) / half_dim))
    angles = torch.arange(seq_len, device=device, dtype=dtype)[:, None] * freqs[None, :]
    return torch.cos(angles), torch.sin(angles)


def apply_rope_separate(q, k, cos, sin):
    """The rotation alone, (q_rot, k_rot), half-split pairs (triton_flash.py:680-703: the table is cut to
    q's sequence length and applied to both).  ROCm tensors run the HIP rotation pass."""
    if not (q.is_cuda and k.is_cuda):
        raise AuleError("aule (HIP build): apply_rope_separate needs ROCm device tensors; there is no CPU fallback")
    from ._torch import _rope_tables, rope_raw
    if k.shape[2] != q.shape[2]:
        raise ValueError(f"apply_rope_separate needs equal sequence lengths, got {q.shape[2]} and {k.shape[2]}")
    c, s = _rope_tables(cos, sin, q.shape[-1], q.device)
    return rope_raw(q.contiguous(), c, s, "half"), rope_raw(k.contiguous(), c, s, "half")


# =============================================================================
# PyTorch SDPA compatibility layer (reference __init__.py:288-442)
# =============================================================================
_original_sdpa = None
_installed = False
_SDPA_DTYPES = ("torch.float16", "torch.bfloat16", "torch.float32")


def scaled_dot_product_attention(query, key, value, attn_mask=None, dropout_p=0.0, is_causal=False,
                                 scale=None, enable_gqa=False):
    """Drop-in for torch.nn.functional.scaled_dot_product_attention (same signature as the reference's
    shim, __init__.py:288-297).  Runs the HIP kernels when it can and defers to PyTorch's own SDPA for
    what the kernels do not cover: attn_mask, dropout, head_dim > 128, non-4-D or non-ROCm tensors,
    dtypes other than fp16/bf16/fp32, and mismatched head counts without enable_gqa (PyTorch raises).
    torch's is_causal mask is top-left aligned, like this library's."""
    import torch
    import torch.nn.functional as F
    fallback = (
        attn_mask is not None or dropout_p > 0.0
        or not (isinstance(query, torch.Tensor) and query.is_cuda and key.is_cuda and value.is_cuda)
        or query.dim() != 4 or key.dim() != 4 or value.dim() != 4
        or query.shape[-1] > 128 or key.shape[-1] != query.shape[-1] or value.shape[-1] != query.shape[-1]
        or str(query.dtype) not in _SDPA_DTYPES
        or (query.shape[1] != key.shape[1] and not enable_gqa)
        or key.shape[1] == 0 or query.shape[1] % max(1, key.shape[1]) != 0
        or key.shape[2] == 0
    )
    if fallback:
        fn = _original_sdpa if _original_sdpa is not None else F.scaled_dot_product_attention
        if fn is scaled_dot_product_attention:   # installed without a saved original: cannot recurse
            raise AuleError("scaled_dot_product_attention fallback requested but the original SDPA is unavailable")
        return fn(query, key, value, attn_mask=attn_mask, dropout_p=dropout_p, is_causal=is_causal,
                  scale=scale, enable_gqa=enable_gqa)
    return flash_attention(query, key, value, causal=is_causal, scale=scale)


def install(backend=None, verbose=False):
    """Route every torch.nn.functional.scaled_dot_product_attention call through this library
    (reference __init__.py:353-406).  `backend` may be None or 'hip' (the only backend of this build)."""
    global _original_sdpa, _installed, _verbose
    import torch
    import torch.nn.functional as F
    if backend is not None and backend != "hip":
        raise ValueError(f"Invalid backend '{backend}'. This build has one backend: 'hip' (or None)")
    _verbose = bool(verbose)
    if _installed:
        print(f"aule-attention: Updated (backend=hip, verbose={verbose})")
        return
    _original_sdpa = F.scaled_dot_product_attention
    F.scaled_dot_product_attention = scaled_dot_product_attention
    torch.nn.functional.scaled_dot_product_attention = scaled_dot_product_attention
    _installed = True
    print("aule-attention: Installed (HIP gfx950%s)" % (", verbose" if verbose else ""))


def uninstall():
    """Restore PyTorch's own SDPA (reference __init__.py:409-430)."""
    global _installed
    if not _installed:
        print("aule-attention: Not installed")
        return
    import torch
    import torch.nn.functional as F
    if _original_sdpa is not None:
        F.scaled_dot_product_attention = _original_sdpa
        torch.nn.functional.scaled_dot_product_attention = _original_sdpa
    _installed = False
    print("aule-attention: Uninstalled, restored PyTorch SDPA")


def get_available_backends():
    """Reference API (__init__.py:445-457); this build has exactly one backend."""
    from . import _capi
    try:
        _capi.get_lib()
        return ["hip"]
    except AuleError:
        return []


def get_backend_errors():
    from . import _capi
    try:
        _capi.get_lib()
        return {}
    except AuleError as e:
        return {"hip": str(e)}


def get_backend_info():
    from . import _capi
    info = {"backends": get_available_backends(), "version": __version__}
    if info["backends"]:
        from .hip import Aule
        info["hip"] = Aule().get_device_info()
        info["library"] = _capi.library_path()
    return info


def print_backend_info():
    """Backend status report (reference: python/aule/__init__.py:516-562); this build lists its one backend, the library it
    loaded and the device the C-ABI reports."""
    print("=" * 60)
    print("AULE-ATTENTION v" + __version__)
    print("=" * 60)
    print()
    backends = get_available_backends()
    print(f"Available backends: {backends}")
    print()
    if backends:
        info = get_backend_info()
        dev = info.get("hip", {})
        print("[1] HIP (gfx950 / MI355X, hand-written kernels behind libaule.so)")
        print(f"    GPU: {dev.get('device_name', 'Unknown')}")
        print(f"    Library: {info.get('library')}")
        print("    Status: FlashAttention-2 forward / backward, fp32 / fp16 / bf16")
        print()
    for name, err in get_backend_errors().items():
        print(f"[-] {name.upper()}: unavailable -- {err}")
        print()
    print("=" * 60)


def set_verbose(flag=True):
    global _verbose
    _verbose = bool(flag)


def __getattr__(name):
    # Aule / GpuTensor: the C-ABI consumer classes of the reference's vulkan.py, exported at package level like the reference does
    # (python/aule/__init__.py:565-592).  Resolved lazily: importing them loads libaule.so, which needs a HIP device.
    if name in ("Aule", "GpuTensor"):
        from . import hip
        return getattr(hip, name)
    raise AttributeError(f"module {__name__!r} has no attribute {name!r}")


__all__ = ["flash_attention", "attention", "flash_attention_paged_amd", "flash_attention_paged",
           "flash_attention_rope", "precompute_rope_frequencies", "apply_rope_separate", "AuleError", "scaled_dot_product_attention", "install", "uninstall",
           "get_available_backends", "get_backend_errors", "get_backend_info", "print_backend_info", "Aule", "GpuTensor", "set_verbose",
           "__version__"]
