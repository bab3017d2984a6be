"""hipGraph capture (torch.cuda.graph) of the forward paths, and the optional caller workspace of the two-launch
short-query paths (abi2: aule_attn_desc.workspace / aule_paged_desc.workspace + the *_workspace_size queries).

A decode step belongs in a graph.  The two-launch paths need a buffer for their partials; allocated inside the library
(hipMallocAsync / hipFreeAsync) it becomes graph nodes that cost more than the kernels of a small step
(tools/graph_check.py: C5b 41 us per replayed step against 18 us eager), so the torch wrapper passes a buffer from
torch's graph-aware caching allocator.  Checked here: capture + replay is bit-identical to eager on every route, the
size query matches what the paths use, and a missing or too small workspace falls back to the library's own allocation
with the same result."""
import ctypes
import math
import os

import pytest

pytestmark = pytest.mark.gpu

SHAPES = [  # name, B, Hq, Hkv, Sq, Sk, D, dtype, causal
    ("route5-c5b", 1, 32, 1, 1, 16384, 64, "fp16", False),
    ("route5-c5c", 1, 32, 1, 64, 16384, 64, "fp16", False),
    ("route4-decode", 8, 32, 8, 1, 8192, 128, "bf16", False),
    ("route5-bottom-right", 2, 16, 4, 48, 4096, 128, "bf16", "bottom-right"),
    ("plain-causal", 1, 8, 8, 1024, 1024, 128, "bf16", True),
    ("route7-causal-split", 1, 8, 8, 4096, 4096, 128, "bf16", True),      # forward over key-range pieces + merge kernel, caller workspace
    ("route7-noncausal-split", 1, 8, 8, 2048, 2048, 128, "fp16", False),
    ("route8-one-wave-per-simd", 4, 16, 16, 1024, 1024, 128, "bf16", True),   # the D = 128 default: several parts per workgroup
    ("route8-fused-rope", 4, 16, 16, 1024, 1024, 128, "bf16", True),      # the forward that rotates Q itself (tables captured too)
    ("route8-window-256", 2, 16, 4, 2048, 2048, 128, "bf16", True),       # round 6: the window instances (late waves, several parts per workgroup)
]


def _mk(torch, B, Hq, Hkv, Sq, Sk, D, dtype):
    dt = {"fp16": torch.float16, "bf16": torch.bfloat16}[dtype]
    g = torch.Generator(device="cuda").manual_seed(11)
    q = torch.randn(B, Hq, Sq, D, device="cuda", dtype=dt, generator=g)
    k = torch.randn(B, Hkv, Sk, D, device="cuda", dtype=dt, generator=g)
    v = torch.randn(B, Hkv, Sk, D, device="cuda", dtype=dt, generator=g)
    return q, k, v


def _capture(torch, fn, steps=4):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2):
            fn()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    outs = []
    with torch.cuda.graph(g):
        for _ in range(steps):
            outs.append(fn())
    return g, outs


@pytest.mark.parametrize("shape", SHAPES, ids=lambda s: s[0])
def test_forward_capture_replays_bit_identical(shape):
    import torch
    from aule import _torch as at
    _, B, Hq, Hkv, Sq, Sk, D, dtype, causal = shape
    q, k, v = _mk(torch, B, Hq, Hkv, Sq, Sk, D, dtype)
    sc = 1 / math.sqrt(D)
    rope = None
    if shape[0].endswith("fused-rope"):
        import aule
        cos, sin = aule.precompute_rope_frequencies(Sk, D)
        rope = (cos.contiguous(), sin.contiguous(), 0)
        assert at.rope_fusable(q, k, 1, -1, rope[0], rope[1], 0)
    W = int(shape[0].rsplit("-", 1)[1]) if "-window-" in shape[0] else -1
    eager, eager_lse = at.fwd_raw(q, k, v, causal, sc, q_rope=rope, window=W)
    torch.cuda.synchronize()
    g, outs = _capture(torch, lambda: at.fwd_raw(q, k, v, causal, sc, q_rope=rope, window=W))
    for o, l in outs:
        o.zero_(); l.zero_()
    g.replay()
    torch.cuda.synchronize()
    for o, l in outs:
        assert torch.equal(o, eager) and torch.equal(l, eager_lse)
    # new inputs in the captured buffers: the replay computes on them (no stale pointers into a freed workspace)
    q.copy_(torch.randn_like(q))
    want, _ = at.fwd_raw(q, k, v, causal, sc, q_rope=rope, window=W)
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(outs[-1][0], want)


def test_paged_decode_capture_replays_bit_identical():
    import torch
    import aule
    torch.manual_seed(3)
    B, Hq, Hkv, D, bs, n = 4, 32, 8, 128, 16, 4096
    nb = n // bs
    kc = torch.randn(B * nb, bs, Hkv, D, device="cuda", dtype=torch.float16)
    vc = torch.randn(B * nb, bs, Hkv, D, device="cuda", dtype=torch.float16)
    q = torch.randn(B, Hq, D, device="cuda", dtype=torch.float16)
    bt = torch.randperm(B * nb, device="cuda").to(torch.int32).view(B, nb)
    cl = torch.tensor([n, 1000, 37, n - 1], device="cuda", dtype=torch.int32)
    eager = aule.flash_attention_paged_amd(q, kc, vc, bt, cl)
    torch.cuda.synchronize()
    g, outs = _capture(torch, lambda: aule.flash_attention_paged_amd(q, kc, vc, bt, cl))
    for o in outs:
        o.zero_()
    g.replay()
    torch.cuda.synchronize()
    assert all(torch.equal(o, eager) for o in outs)


def _desc(torch, q, k, v, out, lse, causal_code):
    from aule import _capi
    B, Hq, Sq, D = q.shape
    d = _capi.AttnDesc()
    d.struct_size = ctypes.sizeof(_capi.AttnDesc)
    d.dtype = {torch.float16: 1, torch.bfloat16: 2}[q.dtype]
    d.batch, d.heads_q, d.heads_kv, d.seq_q, d.seq_k, d.head_dim = B, Hq, k.shape[1], Sq, k.shape[2], D
    d.scale, d.causal, d.window_size, d.device = 1 / math.sqrt(D), causal_code, -1, q.device.index or 0
    d.stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    d.q, d.k, d.v, d.out, d.lse = q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), lse.data_ptr()
    return d


@pytest.mark.parametrize("shape", SHAPES[:4], ids=lambda s: s[0])
def test_caller_workspace_and_fallback_agree(shape):
    """C-ABI: the same problem with (a) the workspace the size query asks for, (b) none, (c) one byte too few
    (must be ignored, not overrun) -- three bit-identical results; and the query is 0 for single-launch paths."""
    import torch
    from aule import _capi
    _, B, Hq, Hkv, Sq, Sk, D, dtype, causal = shape
    q, k, v = _mk(torch, B, Hq, Hkv, Sq, Sk, D, dtype)
    lib = _capi.get_lib()
    code = {False: 0, True: 1, "bottom-right": 2}[causal]
    res = []
    for mode in ("exact", "none", "short"):
        out = torch.empty_like(q)
        lse = torch.empty(B, Hq, Sq, device="cuda", dtype=torch.float32)
        d = _desc(torch, q, k, v, out, lse, code)
        need = int(lib.aule_attention_forward_workspace_size(ctypes.byref(d)))
        assert need > 0
        guard = 4096
        buf = torch.full((need + guard,), 0x5A, device="cuda", dtype=torch.uint8)
        if mode == "exact":
            d.workspace, d.workspace_bytes = buf.data_ptr(), need
        elif mode == "short":
            d.workspace, d.workspace_bytes = buf.data_ptr(), need - 1
        assert lib.aule_attention_forward_ex(ctypes.byref(d)) == 0, lib.aule_get_error()
        torch.cuda.synchronize()
        assert bool((buf[need:] == 0x5A).all()), "wrote past the workspace it was given"
        if mode == "short":
            assert bool((buf[:need] == 0x5A).all()), "used a workspace that was too small"
        res.append((out, lse))
    for o, l in res[1:]:
        assert torch.equal(o, res[0][0]) and torch.equal(l, res[0][1])
    # single-launch path: nothing to provide
    q2, k2, v2 = _mk(torch, 1, 8, 8, 1024, 1024, 128, "bf16")
    d2 = _desc(torch, q2, k2, v2, torch.empty_like(q2), torch.empty(1, 8, 1024, device="cuda"), 1)
    assert int(lib.aule_attention_forward_workspace_size(ctypes.byref(d2))) == 0
