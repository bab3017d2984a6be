"""Sliding window on the GPU: key j is visible to query i only if i - j < W, on top of the
causal rule (the convention of the kernel the reference runs on ROCm, triton_flash_amd.py:179-183).

  * golden vectors recorded from that kernel (tests/golden/win_*.npz, gen_golden.py window), fp32;
  * forward AND backward against the windowed fp64 oracle for 16-bit and fp32 I/O (the reference's own
    backward ignores the window, so there is no reference golden for the gradients);
  * size-independent properties: W >= Sk is full attention bit for bit; W = 1 under a causal mask makes every
    row attend to itself only (O = V, dQ = dK = 0 to rounding, dV = dO summed over the group);
  * the C-ABI handle path (aule_attention_forward_gpu with window_size).
"""
import math
import os

import numpy as np
import pytest

from conftest import golden_files, load_golden
from util import BWD_TOL, LSE_TOL, assert_close, fwd_tol, quantize, torch_dtype

pytestmark = pytest.mark.gpu


def _dev(torch, a, dtype):
    return torch.from_numpy(np.ascontiguousarray(a)).to("cuda", torch_dtype(dtype))


@pytest.mark.parametrize("path", golden_files("win_"), ids=lambda p: p.split("/")[-1][:-4])
def test_window_goldens(path):
    import torch
    import aule
    g = load_golden(path)
    W = int(np.load(path)["window"])
    out = aule.flash_attention(g["q"], g["k"], g["v"], causal=g["causal"], window_size=W)       # numpy in/out
    assert isinstance(out, np.ndarray)
    assert_close(out, g["out"], 1e-5, 1e-5, g["name"])
    from aule import _torch as at
    D = g["q"].shape[-1]
    o, lse = at.fwd_raw(_dev(torch, g["q"], "fp32"), _dev(torch, g["k"], "fp32"), _dev(torch, g["v"], "fp32"),
                        g["causal"], 1 / math.sqrt(D), window=W)
    assert_close(o.cpu().numpy(), g["out"], 1e-5, 1e-5, g["name"] + " out")
    assert_close(lse.cpu().numpy(), g["lse"], 1e-5, 1e-5, g["name"] + " lse")


CASES = [  # dtype, B, Hq, Hkv, Sq, Sk, D, causal, window
    ("bf16", 1, 4, 2, 512, 512, 128, True, 100),
    ("bf16", 1, 2, 2, 1000, 1000, 128, True, 300),
    ("bf16", 1, 2, 1, 333, 500, 64, False, 77),
    ("fp16", 1, 2, 2, 700, 700, 64, True, 64),
    ("bf16", 2, 2, 2, 2048, 2048, 128, True, 512),
    ("fp32", 1, 2, 2, 300, 300, 32, True, 50),
    ("bf16", 1, 2, 2, 600, 200, 128, True, 64),     # rows beyond Sk + W - 1 see no key: O = 0
    ("fp16", 1, 3, 3, 130, 130, 32, True, 7),
    ("bf16", 1, 8, 1, 257, 257, 128, True, 255),
    # round 5: shapes whose grids put the backward on the one-wave-per-SIMD pair by themselves (the mode test of tests/test_gpu_bwd.py forces it
    # onto all of the above as well): several Q blocks and KV block pairs, the window shorter and longer than a block, GQA, D = 64, bottom-right
    ("bf16", 2, 32, 32, 1024, 1024, 128, True, 256),
    ("bf16", 1, 32, 8, 1100, 1100, 128, True, 100),
    ("fp16", 2, 32, 32, 1024, 1024, 64, True, 300),
    ("bf16", 1, 32, 32, 700, 1500, 128, "bottom-right", 200),
    # round 6: the sliding-window instances of the one-wave-per-SIMD forward (causal, W >= 128, every query's diagonal key inside Sk): aligned and
    # unaligned windows, shorter and longer than a Q block, waves that start late (W = 256: wave 3 of a block at its part's tile 3 -> position 2),
    # ragged last blocks, GQA, D = 64, a bottom-right offset that is not a multiple of the tile
    ("bf16", 1, 4, 4, 2048, 2048, 128, True, 256),
    ("bf16", 1, 2, 2, 1536, 1536, 128, True, 128),
    ("bf16", 1, 2, 2, 1536, 1536, 128, True, 64),       # (shorter than two key tiles: the ping-pong kernel)
    ("bf16", 1, 4, 1, 1300, 1300, 64, True, 200),
    ("bf16", 1, 2, 2, 900, 2000, 128, "bottom-right", 700),
    ("bf16", 1, 2, 2, 4096, 4096, 128, True, 1024),
    ("fp16", 1, 2, 2, 1024, 1024, 128, True, 129),
    ("fp16", 1, 4, 2, 3000, 3000, 64, True, 1000),
]
W4_WINDOW_CASES = {c for c in CASES if c[0] != "fp32" and c[6] in (64, 128) and c[7] and c[8] >= 128 and min(256 + (c[5] - c[4] if c[7] == "bottom-right" else 0), c[5]) > 192
                   and c[4] + (c[5] - c[4] if c[7] == "bottom-right" else 0) <= c[5]}


@pytest.mark.parametrize("case", CASES, ids=lambda c: "-".join(str(x) for x in c))
def test_window_forward_backward_vs_oracle(case, oracle_mod):
    import torch
    from aule import _torch as at
    dtype, B, Hq, Hkv, Sq, Sk, D, causal, W = case
    rng = np.random.RandomState(11)
    q, k, v, do = (quantize(rng.randn(*s).astype(np.float32), dtype)
                   for s in ((B, Hq, Sq, D), (B, Hkv, Sk, D), (B, Hkv, Sk, D), (B, Hq, Sq, D)))
    sc = 1 / math.sqrt(D)
    tq, tk, tv, tdo = (_dev(torch, x, dtype) for x in (q, k, v, do))
    out, lse = at.fwd_raw(tq, tk, tv, causal, sc, window=W)
    if not os.environ.get("AULE_HIP_W4_WINDOW") and not os.environ.get("AULE_HIP_FWD_KERNEL") and not os.environ.get("AULE_HIP_FWD_SOFTMAX"):
        assert (_fwd_route(dtype, B, Hq, Hkv, Sq, Sk, D, causal, W) == 8) == (case in W4_WINDOW_CASES), "route"
    ref, ref_lse = oracle_mod.fwd_f64(q, k, v, causal, None, W)
    atol, rtol = fwd_tol(dtype, np.abs(v).max())
    assert_close(out.float().cpu().numpy(), ref, atol, rtol, "out")
    fin = np.isfinite(ref_lse)
    got_lse = lse.cpu().numpy()
    assert_close(got_lse[fin], ref_lse[fin], LSE_TOL[dtype], 1e-5, "lse")
    assert np.all(np.isneginf(got_lse[~fin]))          # rows without a visible key
    dq, dk, dv = at.bwd_raw(tq, tk, tv, out, tdo, lse, causal, sc, window=W)
    rq, rk, rv = oracle_mod.bwd_f64(q, k, v, do, causal, None, W)
    a, r = BWD_TOL[dtype]
    for name, got, want in (("dq", dq, rq), ("dk", dk, rk), ("dv", dv, rv)):
        assert_close(got.float().cpu().numpy(), want, a * max(1.0, float(np.abs(want).max())), r, name)


def _fwd_route(dtype, B, Hq, Hkv, Sq, Sk, D, causal, W):
    """aule_hip_debug_forward_route for the problem (host logic only): 8 = the one-wave-per-SIMD forward, 1 = the ping-pong kernel"""
    import ctypes
    from aule import _capi
    d = _capi.AttnDesc()
    d.struct_size = ctypes.sizeof(_capi.AttnDesc)
    d.batch, d.heads_q, d.heads_kv, d.seq_q, d.seq_k, d.head_dim = B, Hq, Hkv, Sq, Sk, D
    d.dtype = {"fp32": 0, "fp16": 1, "bf16": 2}[dtype]
    d.causal = 2 if causal == "bottom-right" else int(bool(causal))
    d.window_size = W
    return int(_capi.get_lib().aule_hip_debug_forward_route(ctypes.byref(d)))


@pytest.mark.parametrize("dtype,mag", [("bf16", 40.0), ("fp16", 20.0), ("fp16", 3.0)], ids=["bf16-x40", "fp16-x20", "fp16-x3"])
def test_window_with_large_logits_takes_the_exact_maximum_stream(oracle_mod, dtype, mag):
    """The window instances take a row's fixed reference from the wave's first tile under the CAUSAL mask only -- keys in front of the window
    included.  A key just OUTSIDE a block's window whose logit is far above everything inside must not poison the rows (the row sums underflow,
    the range verdict fails, the exact-maximum stream -- which takes the maximum under BOTH bounds -- repairs the part), and neither must a key
    INSIDE the window far above the first tile's scores (overflow).  fp16 (found by tools/fuzz_parity.py window, round 6): a reference a few
    dozen log2 units too HIGH does not underflow the fp32 sums, but pushes every weight below fp16's normal range -- the fp16 window instances
    therefore send every row whose sum of weights is below 1 to the exact-maximum stream (x20: the spiked key is +-29 log2 units for the
    OTHER rows of its head; x3: a handful of units -- weights stay normal, nothing may be repaired wrongly either)."""
    import torch
    from aule import _torch as at
    rng = np.random.RandomState(5)
    B, H, S, D, W = 1, 4, 2048, 128, 256
    tdt = torch_dtype(dtype)
    mk = lambda *s: torch.from_numpy(rng.randn(*s).astype(np.float32)).to(tdt)
    q, k, v = mk(B, H, S, D), mk(B, H, S, D), mk(B, H, S, D)
    # head 0: key 1290 is outside the window of row 1600 (1600 - 1290 = 310 >= 256) but inside the first tile of its wave; aligned with that row's query
    k[:, 0, 1290, :] = (mag * q[:, 0, 1600, :].float()).to(tdt)
    # head 1: key 1500 is inside the window of rows 1500 .. 1755, far above their first tile's scores for row 1600
    k[:, 1, 1500, :] = (mag * q[:, 1, 1600, :].float()).to(tdt)
    out, lse = at.fwd_raw(q.cuda(), k.cuda(), v.cuda(), True, 1 / math.sqrt(D), window=W)
    torch.cuda.synchronize()
    ref, rl = oracle_mod.fwd_f64(q.float().numpy(), k.float().numpy(), v.float().numpy(), True, None, W)
    o = out.float().cpu().numpy()
    assert not np.isnan(o).any()
    atol, rtol = fwd_tol(dtype, float(v.float().abs().max()))
    assert_close(o, ref, atol, rtol, "out")
    assert_close(lse.cpu().numpy(), rl, LSE_TOL[dtype] * max(1.0, mag / 8.0), 1e-5, "lse")


def test_window_with_a_negative_scale(oracle_mod):
    """Round 6: negative scales run the one-wave-per-SIMD kernel on negated Q fragments -- its window instances included (late waves negate in the
    part prologue like everybody else)."""
    import torch
    from aule import _torch as at
    rng = np.random.RandomState(9)
    B, Hq, Hkv, S, D, W, sc = 1, 4, 2, 1024, 128, 256, -0.15
    q, k, v = (quantize(rng.randn(*s).astype(np.float32), "bf16") for s in ((B, Hq, S, D), (B, Hkv, S, D), (B, Hkv, S, D)))
    assert _fwd_route("bf16", B, Hq, Hkv, S, S, D, True, W) == 8
    out, lse = at.fwd_raw(_dev(torch, q, "bf16"), _dev(torch, k, "bf16"), _dev(torch, v, "bf16"), True, sc, window=W)
    ref, rl = oracle_mod.fwd_f64(q, k, v, True, sc, W)
    atol, rtol = fwd_tol("bf16", np.abs(v).max())
    assert_close(out.float().cpu().numpy(), ref, atol, rtol, "out")
    assert_close(lse.cpu().numpy(), rl, LSE_TOL["bf16"] * 2, 1e-5, "lse")


def test_window_launch_with_two_rounds_of_part_tables(oracle_mod):
    """More work items than the part tables of one workgroup per CU hold (64 each): the grid then has 2 x CUs workgroups.  65 blocks x 256 heads on sampled
    rows against the fp64 judge (tools/win_big_check.py is the same check at B 16 H 32 S 16384 D 128)."""
    import torch
    from aule import _torch as at
    B, Hq, Hkv, S, D, W = 4, 64, 8, 16640, 64, 300
    g = torch.Generator(device="cuda").manual_seed(3)
    q = torch.randn(B, Hq, S, D, device="cuda", dtype=torch.float16, generator=g)
    k, v = (torch.randn(B, Hkv, S, D, device="cuda", dtype=torch.float16, generator=g) for _ in range(2))
    assert _fwd_route("fp16", B, Hq, Hkv, S, S, D, True, W) == 8
    out, lse = at.fwd_raw(q, k, v, True, 1 / math.sqrt(D), window=W)
    torch.cuda.synchronize()
    assert torch.isfinite(out.float()).all() and torch.isfinite(lse).all()
    rows = np.random.RandomState(1).randint(0, B * Hq * S, size=48).astype(np.int64)
    ro, rl = oracle_mod.fwd_rows_f64(q.float().cpu().numpy(), k.float().cpu().numpy(), v.float().cpu().numpy(), rows, True, None, W)
    atol, rtol = fwd_tol("fp16", float(v.float().abs().max()))
    assert_close(out.float().cpu().numpy().reshape(-1, D)[rows], ro, atol, rtol, "out")
    assert_close(lse.cpu().numpy().reshape(-1)[rows], rl, LSE_TOL["fp16"], 1e-5, "lse")


def test_window_suite_on_the_ping_pong_route():
    """The same cases with the window instances off (AULE_HIP_W4_WINDOW=0: read once per process, hence a child): the ping-pong kernel's window
    path still serves non-causal windows, windows shorter than two key tiles and rows without a visible key, and stays the A/B partner of
    tools/window_bench.py."""
    import subprocess
    import sys
    e = dict(os.environ)
    e["AULE_HIP_W4_WINDOW"] = "0"
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-x", "-q", "-m", "gpu", "-k",
                        "forward_backward_vs_oracle or goldens or large_logits"], env=e, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


def test_window_properties():
    import torch
    import aule
    torch.manual_seed(3)
    q = torch.randn(1, 4, 384, 128, device="cuda", dtype=torch.bfloat16)
    k = torch.randn(1, 2, 384, 128, device="cuda", dtype=torch.bfloat16)
    v = torch.randn(1, 2, 384, 128, device="cuda", dtype=torch.bfloat16)
    full = aule.flash_attention(q, k, v, causal=True)
    assert torch.equal(aule.flash_attention(q, k, v, causal=True, window_size=384), full)     # W >= Sk: no effect
    assert torch.equal(aule.flash_attention(q, k, v, causal=True, window_size=10 ** 6), full)
    # W = 1 + causal: every row attends to itself only
    qa, ka, va = (x.clone().requires_grad_(True) for x in (q, k, v))
    o1 = aule.flash_attention(qa, ka, va, causal=True, window_size=1)
    want = v.repeat_interleave(2, dim=1)
    assert torch.equal(o1, want)
    do = torch.randn_like(o1)
    o1.backward(do)
    # dS = p (dP - delta) with p = 1: zero up to the rounding difference between dP and delta = rowsum(O dO)
    assert float(qa.grad.abs().max()) < 1e-4 and float(ka.grad.abs().max()) < 1e-4
    want_dv = do.float().view(1, 2, 2, 384, 128).sum(dim=2)
    assert torch.allclose(va.grad.float(), want_dv, atol=2e-2, rtol=2e-2)


def test_window_through_the_c_abi_handles(oracle_mod):
    from aule.hip import Aule
    rng = np.random.RandomState(2)
    q = rng.randn(1, 4, 96, 64).astype(np.float32)
    k = rng.randn(1, 2, 160, 64).astype(np.float32)
    v = rng.randn(1, 2, 160, 64).astype(np.float32)
    with Aule() as a:
        out = a.attention(q, k, v, causal=True, window_size=40)
    ref, _ = oracle_mod.fwd_f64(q, k, v, True, None, 40)
    assert_close(out, ref, 1e-5, 1e-5, "forward_gpu window")
