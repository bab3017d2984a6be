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


def test_window_not_implemented(small_qkv):
    q, k, v = small_qkv
    with pytest.raises(NotImplementedError):
        aule.flash_attention(q, k, v, window_size=16)


def test_version_and_exports():
    assert aule.__version__.startswith("0.5.0")
    for name in ("flash_attention", "attention", "AuleError", "get_available_backends"):
        assert hasattr(aule, name)
