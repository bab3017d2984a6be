"""CPU tests of the Python boundary aule.flash_attention: signature, defaults and the
ValueError conditions of the reference (python/aule/__init__.py:104-160)."""
import inspect

import numpy as np
import pytest

import aule


def test_signature_matches_reference():
    sig = inspect.signature(aule.flash_attention)
    assert list(sig.parameters) == ["query", "key", "value", "rot_cos", "rot_sin", "causal", "scale", "window_size"]
    p = sig.parameters
    assert p["rot_cos"].default is None and p["rot_sin"].default is None
    assert p["causal"].default is True and p["scale"].default is None and p["window_size"].default == -1
    assert aule.attention is aule.flash_attention          # __init__.py:275


def _arr(*shape):
    return np.zeros(shape, np.float32)


@pytest.mark.parametrize("q,k,v,msg", [
    (_arr(2, 4, 8), _arr(1, 2, 4, 8), _arr(1, 2, 4, 8), "query must be 4D"),
    (_arr(1, 2, 4, 8), _arr(2, 4, 8), _arr(1, 2, 4, 8), "key must be 4D"),
    (_arr(1, 2, 4, 8), _arr(1, 2, 4, 8), _arr(2, 4, 8), "value must be 4D"),
    (_arr(1, 2, 4, 8), _arr(2, 2, 4, 8), _arr(2, 2, 4, 8), "Batch size mismatch"),
    (_arr(1, 2, 4, 8), _arr(1, 2, 4, 16), _arr(1, 2, 4, 16), "head_dim mismatch"),
    (_arr(1, 2, 4, 8), _arr(1, 2, 4, 8), _arr(1, 2, 6, 8), "Key/value seq_len mismatch"),
    (_arr(1, 2, 4, 8), _arr(1, 2, 4, 8), _arr(1, 1, 4, 8), "Key/value heads mismatch"),
    (_arr(1, 3, 4, 8), _arr(1, 2, 4, 8), _arr(1, 2, 4, 8), "must be divisible by heads_kv"),
])
def test_validation_errors(q, k, v, msg):
    with pytest.raises(ValueError, match=msg):
        aule.flash_attention(q, k, v)


def test_validation_applies_to_torch_too():
    import torch
    with pytest.raises(ValueError, match="query must be 4D"):
        aule.flash_attention(torch.zeros(2, 3, 4), torch.zeros(1, 2, 3, 4), torch.zeros(1, 2, 3, 4))


def test_window_is_accepted_and_reaches_the_backend(small_qkv):
    """window_size > 0 is a supported option: it passes validation and, on a box without a
    ROCm device, fails like every other call -- with AuleError, not NotImplementedError."""
    import torch
    q, k, v = small_qkv
    if torch.cuda.is_available():
        out = aule.flash_attention(q, k, v, window_size=16)
        assert out.shape == q.shape
    else:
        with pytest.raises(aule.AuleError):
            aule.flash_attention(q, k, v, window_size=16)


def test_version_and_exports(capsys):
    assert aule.__version__.startswith("0.5.0")
    for name in ("flash_attention", "attention", "AuleError", "get_available_backends"):
        assert hasattr(aule, name)
    # the reference's public list (python/aule/__init__.py:565-592) minus what is out of scope here (patch_model: model patching)
    for name in ("flash_attention", "attention", "scaled_dot_product_attention", "flash_attention_rope", "precompute_rope_frequencies",
                 "apply_rope_separate", "flash_attention_paged_amd", "install", "uninstall", "get_available_backends", "get_backend_errors",
                 "get_backend_info", "print_backend_info", "Aule", "GpuTensor", "AuleError", "__version__"):
        assert name in aule.__all__ and getattr(aule, name) is not None, name
    from aule import hip
    assert aule.Aule is hip.Aule and aule.GpuTensor is hip.GpuTensor
    aule.print_backend_info()   # works without a device: reports the backend as unavailable, with the reason
    out = capsys.readouterr().out
    assert "AULE-ATTENTION v" in out and "Available backends" in out


def test_sdpa_shim_install_uninstall_and_cpu_fallback(capsys):
    """Reference __init__.py:288-442: install() swaps F.scaled_dot_product_attention, unsupported
    calls (here: CPU tensors, attn_mask, dropout) defer to the saved original, uninstall() restores."""
    import torch
    import torch.nn.functional as F
    orig = F.scaled_dot_product_attention
    sig = inspect.signature(aule.scaled_dot_product_attention)
    assert list(sig.parameters) == ["query", "key", "value", "attn_mask", "dropout_p", "is_causal", "scale", "enable_gqa"]
    with pytest.raises(ValueError):
        aule.install(backend="vulkan")
    aule.install()
    try:
        assert F.scaled_dot_product_attention is aule.scaled_dot_product_attention
        q = torch.randn(1, 2, 8, 16)
        out = F.scaled_dot_product_attention(q, q, q, is_causal=True)       # CPU tensor -> original SDPA
        assert torch.allclose(out, orig(q, q, q, is_causal=True))
        mask = torch.ones(8, 8, dtype=torch.bool).tril()
        assert torch.allclose(F.scaled_dot_product_attention(q, q, q, attn_mask=mask), orig(q, q, q, attn_mask=mask))
        aule.install(verbose=True)   # re-install only updates flags
    finally:
        aule.uninstall()
        aule.set_verbose(False)
    assert F.scaled_dot_product_attention is orig
    aule.uninstall()
    assert "Not installed" in capsys.readouterr().out


def test_causal_argument_codes():
    """`causal` keeps the reference's bool meaning; the strings are the additive alignment option (C-ABI AULE_CAUSAL_*)."""
    from aule._torch import causal_code
    assert [causal_code(x) for x in (False, None, 0, True, 1, "top-left", "bottom-right", 2, "none")] == \
        [0, 0, 0, 1, 1, 1, 2, 2, 0]
    for bad in ("diagonal", 3, -1):
        with pytest.raises(ValueError):
            causal_code(bad)


def test_rope_exports_match_reference_signatures():
    """flash_attention_rope / precompute_rope_frequencies / apply_rope_separate: triton_flash.py:561-570, :644-650,
    :680 (exported by the reference at __init__.py:72-75)."""
    assert list(inspect.signature(aule.flash_attention_rope).parameters) == \
        ["q", "k", "v", "cos", "sin", "causal", "scale", "window_size"]
    p = inspect.signature(aule.flash_attention_rope).parameters
    assert p["causal"].default is True and p["scale"].default is None and p["window_size"].default == -1
    assert list(inspect.signature(aule.precompute_rope_frequencies).parameters) == \
        ["seq_len", "head_dim", "base", "device", "dtype"]
    assert inspect.signature(aule.precompute_rope_frequencies).parameters["base"].default == 10000.0
    assert list(inspect.signature(aule.apply_rope_separate).parameters) == ["q", "k", "cos", "sin"]
    cos, sin = aule.precompute_rope_frequencies(8, 16, device="cpu")
    assert tuple(cos.shape) == (8, 8) and float(cos[0].min()) == 1.0 and float(sin[0].abs().max()) == 0.0
    import torch
    x = torch.zeros(1, 1, 8, 16)
    with pytest.raises(aule.AuleError):          # CPU tensors: no fallback
        aule.flash_attention_rope(x, x, x, cos, sin)


def test_fused_rotation_rule_from_python():
    """aule._torch.rope_fusable is host logic (shapes, dtypes, table geometry): it answers without a GPU, and the way
    flash_attention_rope uses it -- fused only for inference, half-split pairs, un-padded head_dim -- is what DESIGN 3.6 says."""
    import torch
    from aule import _torch as at
    q = torch.zeros(4, 32, 2048, 64, dtype=torch.bfloat16)
    k = torch.zeros(4, 8, 2048, 64, dtype=torch.bfloat16)
    cos, sin = aule.precompute_rope_frequencies(2048, 64, device="cpu")
    cos, sin = cos.contiguous(), sin.contiguous()
    assert at.rope_fusable(q, k, 1, -1, cos, sin, 0)
    assert at.rope_fusable(q, k, 0, -1, cos, sin, 0)
    assert at.rope_fusable(q, k, 1, 128, cos, sin, 0)                        # (round 6) a causal window of >= two key tiles: the window instances rotate Q too
    assert not at.rope_fusable(q, k, 1, 100, cos, sin, 0) and not at.rope_fusable(q, k, 0, 128, cos, sin, 0)   # shorter / non-causal windows: the ping-pong kernel
    assert at.rope_fusable(q, k, 1, -1, cos, sin, 0, 0.11) and at.rope_fusable(q, k, 1, -1, cos, sin, 0, -0.2)   # (round 6) a negative scale runs the same kernel, on negated Q fragments
    assert not at.rope_fusable(q.float(), k.float(), 1, -1, cos, sin, 0)     # fp32 kernel
    assert not at.rope_fusable(q[..., :32].contiguous(), k[..., :32].contiguous(), 1, -1, cos[:, :16].contiguous(), sin[:, :16].contiguous(), 0)
    assert not at.rope_fusable(q, k, 1, -1, cos[:1000], sin[:1000], 0)       # table shorter than the sequence
    assert not at.rope_fusable(q, k, 1, -1, cos, sin, 1)                     # ... or than sequence + offset
    assert not at.rope_fusable(q[:, :, :1], k, 0, -1, cos, sin, 0)           # one query row: a short-query route
    assert not at.rope_fusable(q, k, 1, -1, cos.t().contiguous().t(), sin, 0)    # table rows not contiguous
