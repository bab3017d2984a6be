"""GPU parity tests of the backward hot path (run with -m gpu on an MI355X).

dQ, dK, dV from aule_attention_backward_ex are compared with the reference's recorded
gradients (tests/golden/tr_*), with the fp64 oracle on seeded inputs, and with torch
autograd through a plain fp32 softmax(QK^T)V (config #3: GQA 32q/8kv S=2048 D=128 bf16)."""
import math
import os
import zlib

import numpy as np
import pytest

from conftest import golden_files, load_golden
from util import BWD_TOL, assert_close, fwd_tol, quantize, torch_dtype

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available()
    return torch


def run_fwd_bwd(torch, q, k, v, do, dtype, causal, scale):
    """Through the public API + autograd."""
    import aule
    dt = torch_dtype(dtype)
    tq, tk, tv = (torch.from_numpy(np.ascontiguousarray(x)).to("cuda", dt).requires_grad_(True) for x in (q, k, v))
    out = aule.flash_attention(tq, tk, tv, causal=causal, scale=scale)
    out.backward(torch.from_numpy(np.ascontiguousarray(do)).to("cuda", dt))
    torch.cuda.synchronize()
    assert tq.grad.dtype == dt and tk.grad.shape == tk.shape
    return (out.detach().float().cpu().numpy(), tq.grad.float().cpu().numpy(), tk.grad.float().cpu().numpy(),
            tv.grad.float().cpu().numpy())


def grad_close(got, ref, dtype, what, sharp=False):
    """sharp: a softmax scale well above 1/sqrt(D) (scores of std >= 2 on N(0,1) inputs): the weights are concentrated, P and dS carry their
    16-bit rounding on a few large terms instead of averaging it over many, and bf16 gradients sit at 8e-3 of max|grad| (B1 6q/3kv S97/161 D64
    scale 0.5: measured in round 5) -- such cases keep the reference's own bar, 1e-2 (python/tests/test_triton.py:92-94)."""
    atol, rtol = BWD_TOL[dtype]
    if sharp and dtype == "bf16":
        atol = 1e-2
    scale = max(1.0, float(np.abs(ref).max()))
    err = np.abs(np.asarray(got, dtype=np.float64) - np.asarray(ref, dtype=np.float64))
    # the raw figure next to the bound (pytest -s, or the captured output of a failure)
    print("%s: max|err| %.3e = %.2e of max|grad| %.3g  [bound %.1e * max(1, max|grad|) + %.0e * |ref|]" % (what, err.max(), err.max() / scale, scale, atol, rtol))
    assert_close(got, ref, atol * scale, rtol, what)


@pytest.mark.parametrize("path", [p for p in golden_files("tr_") if "dq" in np.load(p).files],
                         ids=lambda p: p.split("/")[-1][:-4])
def test_reference_gradients(torch_cuda, path):
    g = load_golden(path)
    _, dq, dk, dv = run_fwd_bwd(torch_cuda, g["q"], g["k"], g["v"], g["dout"], g["dtype"], g["causal"], g["scale"])
    grad_close(dq, g["dq"], g["dtype"], g["name"] + " dq")
    grad_close(dk, g["dk"], g["dtype"], g["name"] + " dk")
    grad_close(dv, g["dv"], g["dtype"], g["name"] + " dv")


SWEEP = [
    # dtype, B, Hq, Hkv, Sq, Sk, D, causal, scale
    ("bf16", 1, 4, 4, 256, 256, 128, True, None),
    ("bf16", 2, 8, 2, 320, 320, 128, True, None),
    ("bf16", 1, 8, 1, 200, 333, 128, False, None),
    ("bf16", 1, 4, 2, 130, 70, 128, True, None),
    ("bf16", 1, 4, 2, 70, 300, 128, True, None),      # Sk >> Sq causal: late KV blocks get zero grads
    ("bf16", 1, 2, 2, 1, 1, 128, True, None),
    # shapes for the one-wave-per-SIMD dK/dV kernel (it takes them by itself on full grids of grouped heads; the variant test
    # below forces it on everything it can run): several 128-key blocks and pairs, ragged rows and keys, the GQA loop
    ("bf16", 2, 8, 2, 640, 640, 128, True, None),
    ("bf16", 1, 8, 2, 515, 771, 128, False, None),
    ("fp16", 2, 4, 1, 1000, 1000, 128, True, 0.11),
    ("bf16", 1, 2, 2, 333, 1100, 128, True, None),
    ("bf16", 1, 4, 4, 512, 512, 64, True, None),
    ("bf16", 1, 6, 3, 97, 161, 64, False, 0.5),
    # ... and their D = 64 instances (round 4: another LDS image, half the MFMAs per block, another register map)
    ("bf16", 2, 8, 2, 640, 640, 64, True, None),
    ("bf16", 1, 8, 2, 515, 771, 64, False, None),
    ("fp16", 2, 4, 1, 1000, 1000, 64, True, 0.11),
    ("bf16", 1, 2, 2, 333, 1100, 64, True, None),
    ("bf16", 1, 4, 4, 192, 192, 32, True, None),
    ("fp16", 1, 4, 1, 384, 384, 64, False, None),
    ("fp16", 1, 4, 4, 300, 300, 128, True, None),
    ("fp32", 1, 8, 8, 256, 256, 64, True, None),
    # round 5: fp32 grids that leave most of the chip idle run every block as key / query range pieces + a sum (fa_bwd_f32.hip): the
    # reference's Zig benchmark shape (tests/benchmark_attention.zig:18-21), causal, and a ragged GQA one
    ("fp32", 4, 8, 8, 512, 512, 64, False, None),
    ("fp32", 2, 8, 2, 333, 700, 64, True, None),
    ("fp32", 1, 4, 2, 150, 150, 128, True, None),
    ("fp32", 2, 4, 1, 77, 201, 64, False, 0.7),
    ("fp32", 1, 2, 2, 129, 129, 32, True, -0.3),
]


@pytest.mark.parametrize("case", SWEEP, ids=lambda c: "-".join(str(x) for x in c))
def test_backward_vs_oracle(torch_cuda, oracle_mod, case):
    dtype, B, Hq, Hkv, Sq, Sk, D, causal, scale = case
    rng = np.random.RandomState(zlib.crc32(repr(case).encode()) & 0xFFFF)
    q = quantize(rng.randn(B, Hq, Sq, D), dtype)
    k = quantize(rng.randn(B, Hkv, Sk, D), dtype)
    v = quantize(rng.randn(B, Hkv, Sk, D), dtype)
    do = quantize(rng.randn(B, Hq, Sq, D), dtype)
    _, dq, dk, dv = run_fwd_bwd(torch_cuda, q, k, v, do, dtype, causal, scale)
    rq, rk, rv = oracle_mod.bwd_f64(q, k, v, do, causal, scale)
    sharp = scale is not None and scale * math.sqrt(D) > 2.0
    grad_close(dq, rq, dtype, "dq", sharp)
    grad_close(dk, rk, dtype, "dk", sharp)
    grad_close(dv, rv, dtype, "dv", sharp)


def _torch_ref(torch, q, k, v, causal, scale):
    g = q.shape[1] // k.shape[1]
    kk = k.repeat_interleave(g, dim=1)
    vv = v.repeat_interleave(g, dim=1)
    s = torch.einsum("bhqd,bhkd->bhqk", q, kk) * scale
    if causal:
        Sq, Sk = s.shape[-2:]
        mask = torch.ones(Sq, Sk, dtype=torch.bool, device=q.device).tril()
        s = s.masked_fill(~mask, float("-inf"))
    return torch.einsum("bhqk,bhkd->bhqd", torch.softmax(s, dim=-1), vv)


def test_config3_gqa_torch_autograd_parity(torch_cuda):
    """BASELINE config #3: GQA 32q/8kv S=2048 D=128 bf16 causal fwd+bwd vs torch autograd
    (plain fp32 math on the same bf16-quantised inputs), at the batch bench.py times (B = 4)."""
    import aule
    torch = torch_cuda
    B, Hq, Hkv, S, D = 4, 32, 8, 2048, 128
    gen = torch.Generator(device="cuda").manual_seed(3)
    q, k, v, do = (torch.randn(B, h, S, D, device="cuda", dtype=torch.bfloat16, generator=gen)
                   for h in (Hq, Hkv, Hkv, Hq))
    q1, k1, v1 = (x.clone().requires_grad_(True) for x in (q, k, v))
    out = aule.flash_attention(q1, k1, v1, causal=True)
    out.backward(do)
    q2, k2, v2 = (x.float().requires_grad_(True) for x in (q, k, v))
    ref = _torch_ref(torch, q2, k2, v2, True, 1.0 / math.sqrt(D))
    ref.backward(do.float())
    assert_close(out.detach().float().cpu().numpy(), ref.detach().cpu().numpy(),
                 *fwd_tol("bf16", v.float().abs().max().item()), "C3 out")
    achieved = {}
    for name, a, b in (("dq", q1.grad, q2.grad), ("dk", k1.grad, k2.grad), ("dv", v1.grad, v2.grad)):
        grad_close(a.float().cpu().numpy(), b.cpu().numpy(), "bf16", "C3 " + name)
        achieved[name] = ((a.float() - b).abs().max() / b.abs().max()).item()
    achieved["out"] = (out.detach().float() - ref.detach()).abs().max().item()
    # the bar is 5e-3 of max|grad| + 1e-2 |ref| (tests/util.py, tightened in round 5); what the kernels actually achieve goes on record (pytest -rP / -s)
    print("C3 B=4 achieved: out max|err| %.3e; max|err| / max|grad|: dq %.3e dk %.3e dv %.3e"
          % (achieved["out"], achieved["dq"], achieved["dk"], achieved["dv"]))
    assert max(achieved["dq"], achieved["dk"], achieved["dv"]) < 5e-3   # (round 4 asserted 1e-2; achieved 2.3 .. 3.4e-3)


FULL_SIZE = [
    # name, B, Hq, Hkv, S, D, the (batch, kv-head) units whose gradients are judged
    ("c2", 4, 32, 32, 4096, 128, ((0, 0), (2, 13), (3, 31))),          # BASELINE configs[1] fwd+bwd: bench.py's extra.c2_fwd_bwd_*, the metric's own shape
    ("c4shard", 8, 32, 32, 8192, 128, ((0, 7), (7, 31))),              # configs[3]'s per-GPU shard (B = 64 over 8 GPUs): 64 blocks per stream
    ("c3", 4, 32, 8, 2048, 128, ((1, 0), (3, 7))),                     # configs[2] again, head by head in fp64 (the test above: torch fp32 autograd)
    ("d64", 8, 32, 32, 2048, 64, ((0, 3), (7, 30))),                   # bench.py's D = 64 training shape
]


@pytest.mark.parametrize("case", FULL_SIZE, ids=lambda c: c[0])
def test_gradients_at_the_sizes_the_bench_times(torch_cuda, oracle_mod, case):
    """Gradient parity at full size (replaces python/aule/triton_flash_amd.py:447-500 at the shapes bench.py times):
    dQ, dK, dV of whole (batch, kv-head) units against the per-head fp64 judge oracle.bwd_head_f64 -- the softmax recomputed in fp64
    from Q and K, no saved LSE -- with the suite's bf16 bound, raw errors printed.  The mode test below runs this file under
    AULE_HIP_BWD_MODE=spill and under both kernel generations too."""
    import aule
    torch = torch_cuda
    name, B, Hq, Hkv, S, D, units = case
    g = Hq // Hkv
    gen = torch.Generator(device="cuda").manual_seed(zlib.crc32(name.encode()) & 0xFFFF)
    q, do = (torch.randn(B, Hq, S, D, device="cuda", dtype=torch.bfloat16, generator=gen) for _ in range(2))
    k, v = (torch.randn(B, Hkv, S, D, device="cuda", dtype=torch.bfloat16, generator=gen) for _ in range(2))
    q.requires_grad_(True); k.requires_grad_(True); v.requires_grad_(True)
    aule.flash_attention(q, k, v, causal=True).backward(do)
    torch.cuda.synchronize()
    for (b, hk) in units:
        hs = slice(hk * g, (hk + 1) * g)
        f = lambda t: t.detach().float().cpu().numpy()
        rq, rk, rv = oracle_mod.bwd_head_f64(f(q[b, hs]), f(k[b, hk]), f(v[b, hk]), f(do[b, hs]), None, True)
        grad_close(f(q.grad[b, hs]), rq, "bf16", "%s dq[%d,%d]" % (name, b, hk))
        grad_close(f(k.grad[b, hk]), rk, "bf16", "%s dk[%d,%d]" % (name, b, hk))
        grad_close(f(v.grad[b, hk]), rv, "bf16", "%s dv[%d,%d]" % (name, b, hk))
    if name == "d64" and not any(os.environ.get(k) for k in ("AULE_HIP_BWD_MODE", "AULE_HIP_BWD_DKV", "AULE_HIP_BWD_DQ", "AULE_HIP_BWD_DKV_K2")):
        # bench.py's D = 64 training shape takes the two-key-blocks-per-wave dK/dV instance by itself (1024 work items of 256 keys)
        from aule import _capi
        assert int(_capi.get_lib().aule_hip_debug_last_backward_route()) == 2 | 4 | 64
    # every unit that was not judged in fp64: finite, and not left at its allocation's content (the kernels write every row)
    for t in (q.grad, k.grad, v.grad):
        assert torch.isfinite(t.float()).all()
        assert (t.float().abs().amax(dim=(-1, -2)) > 0).all()


def test_auto_mode_takes_the_5_matmul_backward_for_cache_sized_problems(torch_cuda):
    """Default dispatch (round 5): a problem whose dS fits the Infinity Cache budget (AULE_HIP_BWD_DS_AUTO_MB, 160) and whose dK/dV grid runs the
    one-wave-per-SIMD kernel takes the 5-matmul backward by itself -- B1 H8 S2048 D128: 33 MB of dS; the workspace size shows the mode -- and
    matches torch autograd like every other mode."""
    import ctypes
    import aule
    from aule import _capi
    if os.environ.get("AULE_HIP_BWD_MODE", "")[:1] == "r" or os.environ.get("AULE_HIP_BWD_DKV", "")[:1] == "o":
        pytest.skip("a mode that never spills is pinned in this environment")
    torch = torch_cuda
    B, H, S, D = 1, 8, 2048, 128
    lib = _capi.get_lib()
    d = _capi.AttnBwdDesc()
    d.struct_size = ctypes.sizeof(_capi.AttnBwdDesc)
    d.dtype, d.causal, d.window_size = 2, 1, -1
    d.batch, d.heads_q, d.heads_kv, d.seq_q, d.seq_k, d.head_dim = B, H, H, S, S, D
    ds_bytes = H * (S // 32) * (S // 32) * 2048
    assert int(lib.aule_attention_backward_workspace_size(ctypes.byref(d))) >= ds_bytes
    gen = torch.Generator(device="cuda").manual_seed(11)
    q, k, v, do = (torch.randn(B, H, S, D, device="cuda", dtype=torch.bfloat16, generator=gen) for _ in range(4))
    q1, k1, v1 = (x.clone().requires_grad_(True) for x in (q, k, v))
    aule.flash_attention(q1, k1, v1, causal=True).backward(do)
    torch.cuda.synchronize()
    # the mode actually taken (a silent fall-back to the recompute pair used to pass this test): bit 1 = the 5-matmul mode
    assert int(lib.aule_hip_debug_last_backward_route()) & 1, lib.aule_hip_debug_last_backward_route()
    q2, k2, v2 = (x.float().requires_grad_(True) for x in (q, k, v))
    _torch_ref(torch, q2, k2, v2, True, 1.0 / math.sqrt(D)).backward(do.float())
    for name, a, b in (("dq", q1.grad, q2.grad), ("dk", k1.grad, k2.grad), ("dv", v1.grad, v2.grad)):
        grad_close(a.float().cpu().numpy(), b.cpu().numpy(), "bf16", "auto-mode " + name)
    # ... and a shape past the budget (C3: 537 MB of touched dS) takes the recompute pair on the one-wave-per-SIMD kernels by itself
    if not os.environ.get("AULE_HIP_BWD_MODE") and not os.environ.get("AULE_HIP_BWD_DKV") and not os.environ.get("AULE_HIP_BWD_DQ"):
        qa, ka, va, da = (torch.randn(4, h, 2048, 128, device="cuda", dtype=torch.bfloat16, generator=gen) for h in (32, 8, 8, 32))
        qa.requires_grad_(True); ka.requires_grad_(True); va.requires_grad_(True)
        aule.flash_attention(qa, ka, va, causal=True).backward(da)
        torch.cuda.synchronize()
        assert int(lib.aule_hip_debug_last_backward_route()) == 2 | 4, lib.aule_hip_debug_last_backward_route()


def test_sgd_step_lowers_loss(torch_cuda):
    """Intent of the reference's (dead) tests/test_torch_autograd.py:63: one SGD step lowers an MSE."""
    import aule
    torch = torch_cuda
    torch.manual_seed(0)
    q = torch.randn(1, 4, 64, 64, device="cuda", requires_grad=True)
    k = torch.randn(1, 4, 64, 64, device="cuda", requires_grad=True)
    v = torch.randn(1, 4, 64, 64, device="cuda", requires_grad=True)
    target = torch.randn(1, 4, 64, 64, device="cuda")
    loss0 = ((aule.flash_attention(q, k, v) - target) ** 2).mean()
    loss0.backward()
    with torch.no_grad():
        for t in (q, k, v):
            t -= 0.5 * t.grad
    loss1 = ((aule.flash_attention(q, k, v) - target) ** 2).mean()
    assert loss1.item() < loss0.item()


def test_backward_deterministic(torch_cuda):
    """No atomics anywhere: two runs are bit-identical."""
    import aule
    torch = torch_cuda
    gen = torch.Generator(device="cuda").manual_seed(9)
    q, k, v, do = (torch.randn(2, h, 384, 128, device="cuda", dtype=torch.bfloat16, generator=gen)
                   for h in (8, 2, 2, 8))
    grads = []
    for _ in range(2):
        a, b, c = (x.clone().requires_grad_(True) for x in (q, k, v))
        aule.flash_attention(a, b, c, causal=True).backward(do)
        grads.append((a.grad, b.grad, c.grad))
    for x, y in zip(*grads):
        assert torch.equal(x, y)


@pytest.mark.parametrize("which", ["spill", "new", "old", "k2", "k1"])
def test_backward_on_the_one_wave_per_simd_kernels(which):
    """Three backward modes over the same suites (the switches are read once per process, hence subprocesses):
      spill  AULE_HIP_BWD_MODE=spill: the 5-matmul backward (round 5: delta pass, fa_bwd_dkv4_gfx950.hip's SPILL instances,
             fa_bwd_dqs_gfx950.hip; an opt-in mode, profiles/r5_bwd_spill.txt) forced onto every problem it CAN run
             (AULE_HIP_BWD_DKV=new lifts the grid rule);
      new    AULE_HIP_BWD_MODE=recompute + the one-wave-per-SIMD pair (fa_bwd_dkv4 / fa_bwd_dq4) forced wherever it can run;
      old    AULE_HIP_BWD_MODE=recompute + both predecessors (fa_bwd_gfx950.hip) everywhere;
      k2     (round 6) as `new`, and every D = 64 problem on the dK/dV instance with TWO key blocks per wave (AULE_HIP_BWD_DKV_K2=1: 256-key work
             items; by itself only grids that fill the chip take it) -- its masks per key block, ragged second blocks, the one-buffer delta pipeline;
      k1     as `new` with that instance off: the one-block D = 64 stream on the shapes that would leave it.
    The sweep, the reference's golden gradients, the bottom-right cases and the determinism test then exercise the masks, the stream
    start / tail, the idle waves, the GQA loop and the dS workspace addressing of each mode on all of them."""
    import subprocess
    import sys
    from conftest import ROOT
    e = dict(os.environ)
    if which == "spill":
        e["AULE_HIP_BWD_DKV"] = "new"
        e["AULE_HIP_BWD_MODE"] = "spill"
    elif which in ("k2", "k1"):
        e["AULE_HIP_BWD_MODE"] = "recompute"
        e["AULE_HIP_BWD_DKV"] = e["AULE_HIP_BWD_DQ"] = "new"
        e["AULE_HIP_BWD_DKV_K2"] = "1" if which == "k2" else "0"
    else:
        e["AULE_HIP_BWD_MODE"] = "recompute"
        e["AULE_HIP_BWD_DKV"] = which
        e["AULE_HIP_BWD_DQ"] = which
    # (round 5: the one-wave-per-SIMD pair takes causal sliding windows too -- the window suite rides along in every leg)
    # (round 6: a leg runs what its switches can change -- the forward-only tests of the window file stay out (one of them is a child suite of its
    # own), the k2 / k1 legs keep to the cases with a 64 in their id (every D = 64 case, and some more), the predecessor leg to the backward file: the
    # whole suite had grown from 5 to 10 minutes on the GPU box, 6 of them in these five children and most of that in the CPU judge's repeats)
    files = [os.path.join(ROOT, "tests", "test_gpu_bwd.py")]
    if which != "old":
        files += [os.path.join(ROOT, "tests", "test_gpu_bottom_right.py"), os.path.join(ROOT, "tests", "test_gpu_window.py")]
    sel = "not one_wave_per_simd and not ping_pong_route and not large_logits and not negative_scale and not goldens and not c_abi and not two_rounds"
    if which in ("k2", "k1"):
        sel += " and 64"
    r = subprocess.run([sys.executable, "-m", "pytest"] + files + ["-q", "-x", "-m", "gpu", "-k", sel], env=e, capture_output=True, text=True, timeout=1500, cwd=ROOT)
    assert r.returncode == 0 and "passed" in r.stdout, r.stdout[-3000:] + r.stderr[-1000:]


def test_spill_backward_in_batch_chunks_is_bit_identical(torch_cuda):
    """The dS workspace is used in chunks of as many batch elements as the caller's buffer holds: a buffer for one element (three
    chunks) and the full one give the same bits; a buffer without any dS room runs the recompute pair (close, not identical)."""
    import ctypes
    import aule
    from aule import _capi, _torch as at
    torch = torch_cuda
    lib = _capi.get_lib()
    gen = torch.Generator(device="cuda").manual_seed(21)
    B, Hq, Hkv, S, D = 3, 8, 2, 640, 128
    q, k, v, do = (torch.randn(B, h, S, D, device="cuda", dtype=torch.bfloat16, generator=gen) for h in (Hq, Hkv, Hkv, Hq))
    sc = 1 / math.sqrt(D)
    out, lse = at.fwd_raw(q, k, v, True, sc)

    def run(ws_bytes):
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        d = _capi.AttnBwdDesc()
        d.struct_size = ctypes.sizeof(_capi.AttnBwdDesc)
        d.dtype = 2
        d.batch, d.heads_q, d.heads_kv, d.seq_q, d.seq_k, d.head_dim = B, Hq, Hkv, S, S, D
        d.scale, d.causal, d.window_size, d.device = sc, 1, -1, torch.cuda.current_device()
        d.stream = torch.cuda.current_stream().cuda_stream
        d.q, d.k, d.v, d.out, d.dout, d.lse = (t.data_ptr() for t in (q, k, v, out, do, lse))
        d.dq, d.dk, d.dv = dq.data_ptr(), dk.data_ptr(), dv.data_ptr()
        full = int(lib.aule_attention_backward_workspace_size(ctypes.byref(d)))
        n = full if ws_bytes is None else ws_bytes(full)
        ws = torch.empty((n,), device="cuda", dtype=torch.uint8)
        d.workspace, d.workspace_bytes = ws.data_ptr(), n
        _capi.check(lib.aule_attention_backward_ex(ctypes.byref(d)), "aule_attention_backward_ex")
        torch.cuda.synchronize()
        return full, dq, dk, dv

    full, dq, dk, dv = run(None)
    per_batch = Hkv * 4 * ((S + 127) // 128) * (Hq // Hkv) * ((S + 31) // 32) * 2048
    base = full - B * per_batch
    if base <= 0:      # no dS part: the grid rule keeps this small problem on the recompute pair unless AULE_HIP_BWD_DKV=new (the mode test's
        pytest.skip("no dS workspace for this shape in this mode")   # "spill" leg runs this test with it), or the mode is recompute
    _, dq1, dk1, dv1 = run(lambda f: base + per_batch)     # one element per chunk
    assert torch.equal(dq, dq1) and torch.equal(dk, dk1) and torch.equal(dv, dv1)
    _, dq0, dk0, dv0 = run(lambda f: base)                 # no dS room: the recompute pair
    for a, b in ((dq, dq0), (dk, dk0), (dv, dv0)):
        assert (a.float() - b.float()).abs().max().item() <= 2e-2 * max(1.0, b.float().abs().max().item())


def test_dq_of_the_one_wave_per_simd_kernel_is_bit_identical_to_its_predecessor():
    """fa_bwd_dq4_gfx950.hip keeps its predecessor's accumulation order over the keys, so dQ must come out bit for bit the same -- at
    D = 128 and, since round 4, at D = 64 (another LDS image, another register map).  The switch is read once per process: two
    subprocesses print the SHA-256 of dQ for the same seeded problems (block pairs, ragged rows and keys, GQA, bottom-right, fp16)."""
    import subprocess
    import sys
    from conftest import ROOT
    prog = r'''
import hashlib, math, sys, torch
sys.path.insert(0, sys.argv[1])
from aule import _torch as at
g = torch.Generator(device="cuda").manual_seed(5)
for (dt, B, Hq, Hkv, Sq, Sk, D, causal) in ((torch.bfloat16, 2, 8, 2, 640, 640, 128, True), (torch.bfloat16, 2, 8, 2, 640, 640, 64, True),
                                            (torch.float16, 1, 4, 4, 333, 900, 64, "bottom-right"), (torch.bfloat16, 1, 6, 3, 515, 771, 64, False),
                                            (torch.bfloat16, 4, 16, 16, 1024, 1024, 64, True)):
    q = torch.randn(B, Hq, Sq, D, device="cuda", dtype=dt, generator=g); k = torch.randn(B, Hkv, Sk, D, device="cuda", dtype=dt, generator=g)
    v = torch.randn(B, Hkv, Sk, D, device="cuda", dtype=dt, generator=g); do = torch.randn(B, Hq, Sq, D, device="cuda", dtype=dt, generator=g)
    sc = 1 / math.sqrt(D)
    out, lse = at.fwd_raw(q, k, v, causal, sc)
    dq, dk, dv = at.bwd_raw(q, k, v, out, do, lse, causal, sc)
    torch.cuda.synchronize()
    print("DQ", hashlib.sha256(dq.view(torch.int16).cpu().numpy().tobytes()).hexdigest())
'''
    outs = {}
    for which in ("old", "new"):
        e = dict(os.environ)
        e["AULE_HIP_BWD_DQ"] = which
        e["AULE_HIP_BWD_MODE"] = "recompute"      # (the default 5-matmul backward has no recomputing dQ kernel at all)
        r = subprocess.run([sys.executable, "-c", prog, os.path.join(ROOT, "aule-attention_amd")], env=e, capture_output=True, text=True, timeout=600, cwd=ROOT)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        outs[which] = [ln for ln in r.stdout.splitlines() if ln.startswith("DQ ")]
    assert len(outs["old"]) == 5 and outs["old"] == outs["new"], outs
