"""GPU parity tests of the forward hot path (run with -m gpu on an MI355X).

The HIP path (aule.flash_attention -> libaule.so -> gfx950 kernels) is compared with
  * the golden vectors recorded from the reference (tests/golden/),
  * the fp64 oracle on the same seeded inputs, at sizes the oracle finishes in seconds,
  * at BASELINE.json's full sizes: sampled rows against the oracle plus size-independent
    properties (causal prefix invariance, batch independence, linearity in V).
Tolerances: tests/util.py (1e-5 fp32; 1e-3 + storage ulp for fp16/bf16).
"""
import math
import zlib

import numpy as np
import pytest

from conftest import golden_files, load_golden
from util import FWD_TOL, LSE_TOL, assert_close, assert_close_rows, fwd_tol, quantize, torch_dtype

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch


def run_fwd(torch, q, k, v, dtype, causal, scale, want_lse=True):
    """numpy fp32 (already quantised) -> device tensors of `dtype` -> HIP kernel -> numpy fp32."""
    from aule import _torch as at
    dt = torch_dtype(dtype)
    tq, tk, tv = (torch.from_numpy(np.ascontiguousarray(x)).to("cuda", dt) for x in (q, k, v))
    D = q.shape[-1]
    sc = (1.0 / math.sqrt(D)) if scale is None else scale
    out, lse = at.fwd_raw(tq, tk, tv, causal, sc, want_lse=want_lse)
    torch.cuda.synchronize()
    return out.float().cpu().numpy(), (lse.cpu().numpy() if lse is not None else None)


def test_native_library_is_loaded(torch_cuda):
    import aule
    from aule import _capi
    assert aule.get_available_backends() == ["hip"]
    assert _capi.library_path().endswith("aule-attention_amd/aule/lib/libaule.so")
    maps = open("/proc/self/maps").read()
    assert "libaule.so" in maps
    info = aule.get_backend_info()
    assert info["hip"]["backend"] == "HIP/ROCm" and info["hip"]["subgroup_size"] == 64


# ---------------------------------------------------------------- golden vectors --------
@pytest.mark.parametrize("path", golden_files("np_"), ids=lambda p: p.split("/")[-1][:-4])
def test_numpy_goldens_numpy_in_numpy_out(torch_cuda, path):
    """Config C1 and the small MHA cases: the reference's NumPy route vs ours (fp32 kernels)."""
    import aule
    g = load_golden(path)
    out = aule.flash_attention(g["q"], g["k"], g["v"], causal=g["causal"])
    assert isinstance(out, np.ndarray) and out.dtype == np.float32 and out.shape == g["q"].shape
    assert_close(out, g["out"], 1e-5, 1e-5, g["name"])


@pytest.mark.parametrize("path", golden_files("tr_") + golden_files("amd_"), ids=lambda p: p.split("/")[-1][:-4])
def test_triton_goldens(torch_cuda, path):
    g = load_golden(path)
    out, lse = run_fwd(torch_cuda, g["q"], g["k"], g["v"], g["dtype"], g["causal"], g["scale"])
    # the golden is itself a 16-bit result of the reference kernel -> both sides carry the
    # storage / P-cast rounding (util.fwd_tol)
    atol, rtol = fwd_tol(g["dtype"], np.abs(g["v"]).max(), sides=2 if g["dtype"] != "fp32" else 1)
    assert_close(out, g["out"], atol, rtol, g["name"] + " out")
    assert_close(lse, g["lse"], LSE_TOL[g["dtype"]], 1e-5, g["name"] + " lse")


# ---------------------------------------------------------------- oracle sweep ---------
SWEEP = [
    # dtype, B, Hq, Hkv, Sq, Sk, D, causal, scale
    ("bf16", 1, 4, 4, 256, 256, 128, True, None),
    ("bf16", 2, 8, 2, 320, 320, 128, True, None),      # GQA, partial last Q block + ragged KV tile
    ("bf16", 1, 8, 1, 200, 333, 128, False, None),     # MQA cross-attn, odd lengths
    ("bf16", 1, 4, 2, 130, 70, 128, True, None),       # Sq > Sk, causal top-left
    ("bf16", 1, 2, 2, 1, 1, 128, True, None),
    ("bf16", 1, 2, 1, 1, 513, 128, False, None),       # decode-like
    ("bf16", 1, 4, 4, 512, 512, 64, True, None),
    ("bf16", 1, 6, 3, 97, 161, 64, False, 0.5),
    ("bf16", 1, 4, 4, 192, 192, 32, True, None),
    ("fp16", 1, 4, 1, 384, 384, 64, False, None),      # config C5 family (MQA fp16 non-causal)
    ("fp16", 1, 4, 4, 300, 300, 128, True, None),
    ("fp16", 2, 4, 2, 65, 129, 32, True, 0.2),
    ("fp32", 1, 8, 8, 256, 256, 64, True, None),       # config C1 shape
    # round 5: fp32 grids that leave most of the chip idle run every Q block as key-range pieces + a merge (fa_fwd_f32.hip, f32_pieces):
    # the reference's own Zig benchmark shape (tests/benchmark_attention.zig:18-21), a ragged GQA one, and most of the fp32 cases below
    ("fp32", 4, 8, 8, 512, 512, 64, False, None),
    ("fp32", 2, 8, 2, 333, 700, 64, True, None),
    ("fp32", 1, 4, 4, 1000, 1000, 128, True, None),
    ("fp32", 1, 4, 2, 150, 150, 128, True, None),
    ("fp32", 2, 4, 1, 77, 201, 64, False, 0.7),
    ("fp32", 1, 2, 2, 129, 129, 32, True, None),
    ("fp32", 1, 2, 2, 40, 40, 128, True, -0.3),        # negative scale
    ("bf16", 1, 2, 2, 96, 96, 128, True, -0.2),
    # round 6: negative scales on the one-wave-per-SIMD kernel (Q fragments negated in registers): several parts per workgroup, GQA, ragged, D = 64,
    # a bottom-right offset, a grid small enough for the key-range split
    ("bf16", 2, 8, 2, 1024, 1024, 128, True, -0.2),
    ("fp16", 1, 4, 4, 700, 1300, 64, False, -0.1),
    ("bf16", 1, 4, 1, 900, 2000, 128, "bottom-right", -0.05),
    ("bf16", 1, 8, 8, 4096, 4096, 128, True, -0.09),
    # scale = 0: uniform attention over the visible keys (the reference's and the oracle's reading; the C-ABI's
    # "0 = default" sentinel is resolved above it, aule/_torch.py:_abi_scale)
    ("bf16", 1, 2, 2, 300, 300, 128, True, 0.0),
    ("fp16", 1, 4, 2, 200, 130, 64, False, 0.0),
    ("fp32", 1, 2, 2, 96, 96, 64, True, 0.0),
]


@pytest.mark.parametrize("case", SWEEP, ids=lambda c: "-".join(str(x) for x in c))
def test_forward_vs_oracle(torch_cuda, oracle_mod, case):
    dtype, B, Hq, Hkv, Sq, Sk, D, causal, scale = case
    rng = np.random.RandomState(zlib.crc32(repr(case).encode()) & 0xFFFF)
    q = quantize(rng.randn(B, Hq, Sq, D), dtype)
    k = quantize(rng.randn(B, Hkv, Sk, D), dtype)
    v = quantize(rng.randn(B, Hkv, Sk, D), dtype)
    out, lse = run_fwd(torch_cuda, q, k, v, dtype, causal, scale)
    ref, ref_lse = oracle_mod.fwd_f64(q, k, v, causal, scale)
    atol, rtol = fwd_tol(dtype, np.abs(v).max())
    assert_close(out, ref, atol, rtol, "out")
    assert_close(lse, ref_lse, LSE_TOL[dtype], 1e-5, "lse")


def test_public_api_torch_roundtrip(torch_cuda, oracle_mod):
    """flash_attention(torch cuda) keeps device/dtype/shape; non-contiguous inputs; head_dim 96 (padded)."""
    import aule
    torch = torch_cuda
    rng = np.random.RandomState(11)
    for dtype, D in (("bf16", 96), ("fp16", 80), ("fp32", 48)):
        q = quantize(rng.randn(1, 4, 100, D), dtype)
        k = quantize(rng.randn(1, 2, 120, D), dtype)
        v = quantize(rng.randn(1, 2, 120, D), dtype)
        dt = torch_dtype(dtype)
        # build non-contiguous [B,H,S,D] views of [B,S,H,D] storage
        tq = torch.from_numpy(q).to("cuda", dt).transpose(1, 2).contiguous().transpose(1, 2)
        tk = torch.from_numpy(k).to("cuda", dt).transpose(1, 2).contiguous().transpose(1, 2)
        tv = torch.from_numpy(v).to("cuda", dt)
        assert not tq.is_contiguous()
        out = aule.flash_attention(tq, tk, tv, causal=True)
        assert out.device == tq.device and out.dtype == dt and tuple(out.shape) == tuple(tq.shape)
        ref, _ = oracle_mod.fwd_f64(q, k, v, True, None)
        atol, rtol = fwd_tol(dtype, np.abs(v).max())
        assert_close(out.float().cpu().numpy(), ref, atol, rtol, f"{dtype} D={D}")
    # fp64 input is computed in fp32 and cast back (reference: "other dtypes -> fp32", triton_flash.py:405-411)
    q64 = torch.randn(1, 2, 33, 64, device="cuda", dtype=torch.float64)
    o64 = aule.flash_attention(q64, q64, q64)
    assert o64.dtype == torch.float64
    # CPU torch tensor comes back on the CPU
    qc = torch.randn(1, 2, 16, 32)
    oc = aule.flash_attention(qc, qc, qc, causal=False)
    assert oc.device.type == "cpu" and tuple(oc.shape) == (1, 2, 16, 32)
    # empty batch
    e = aule.flash_attention(torch.empty(0, 2, 8, 64, device="cuda"), torch.empty(0, 2, 8, 64, device="cuda"),
                             torch.empty(0, 2, 8, 64, device="cuda"))
    assert tuple(e.shape) == (0, 2, 8, 64)


def test_rope_args_warn_and_are_ignored(torch_cuda):
    import aule
    torch = torch_cuda
    q = torch.randn(1, 2, 32, 64, device="cuda", dtype=torch.bfloat16)
    base = aule.flash_attention(q, q, q)
    with pytest.warns(UserWarning):
        with_rope = aule.flash_attention(q, q, q, torch.ones(32, 32), torch.zeros(32, 32))
    assert torch.equal(base, with_rope)


def test_online_softmax_rescale_is_exercised(torch_cuda, oracle_mod):
    """A late key with a huge logit forces the running max to jump in the LAST tile
    (cdna guide rule: a rare data-dependent branch needs its own input)."""
    rng = np.random.RandomState(5)
    for dtype in ("bf16", "fp32"):
        q = rng.randn(1, 2, 300, 128).astype(np.float32) * 0.5
        k = rng.randn(1, 2, 300, 128).astype(np.float32) * 0.5
        v = rng.randn(1, 2, 300, 128).astype(np.float32)
        k[:, :, 290, :] = 6.0 * q[:, :, 295, :]        # spike: row 295 (and others) vs key 290
        k[:, :, 100, :] = -4.0 * q[:, :, 120, :]
        q, k, v = (quantize(x, dtype) for x in (q, k, v))
        for causal in (True, False):
            out, lse = run_fwd(torch_cuda, q, k, v, dtype, causal, None)
            ref, ref_lse = oracle_mod.fwd_f64(q, k, v, causal, None)
            atol, rtol = fwd_tol(dtype, np.abs(v).max())
            assert_close(out, ref, atol, rtol, f"spike {dtype} causal={causal}")
            assert_close(lse, ref_lse, LSE_TOL[dtype] * 4, 1e-5, "spike lse")


def test_fp16_fixed_reference_verdict(torch_cuda, oracle_mod):
    """fp16 runs the fixed-reference softmax with a verdict that must catch weights beyond 65504: (a) logits far above the
    first tile's maximum -> the Q block is recomputed with the online form; (b) 33 024 equal logits per row -> row sums
    beyond the 2^15 verdict bound without any overflow: recomputed, and exact either way."""
    from aule import _torch as at
    torch = torch_cuda
    rng = np.random.RandomState(11)
    q, k, v = (quantize(rng.randn(1, 2, 512, 128) * 6.0, "fp16") for _ in range(3))
    out, lse = run_fwd(torch, q, k, v, "fp16", True, None)
    ref, ref_lse = oracle_mod.fwd_f64(q, k, v, True, None)
    assert_close(out, ref, *fwd_tol("fp16", np.abs(v).max()), "fp16 large logits")
    assert_close(lse, ref_lse, LSE_TOL["fp16"] * 6, 1e-5, "fp16 large logits lse")
    B, H, Sq, Sk, D = 16, 16, 512, 33024, 64
    gen = torch.Generator(device="cuda").manual_seed(12)
    qz = torch.zeros(B, H, Sq, D, device="cuda", dtype=torch.float16)
    kz, vz = (torch.randn(B, H, Sk, D, device="cuda", dtype=torch.float16, generator=gen) for _ in range(2))
    o, l = at.fwd_raw(qz, kz, vz, False, 0.125)
    torch.cuda.synchronize()
    mean_v = vz.float().mean(dim=2, keepdim=True)            # uniform attention = the mean of V
    assert (o.float() - mean_v).abs().max().item() < 1e-3
    assert (l - math.log(Sk)).abs().max().item() < 1e-4


# ---------------------------------------------------------------- full-size configs -----
def _sample_rows(rng, total, n):
    return np.unique(np.concatenate([[0, total - 1], rng.randint(0, total, size=n)])).astype(np.int64)


def test_config2_full_size_sampled_rows_and_properties(torch_cuda, oracle_mod):
    """BASELINE config #2: B4 H32 S4096 D128 bf16 causal, fwd."""
    import aule
    torch = torch_cuda
    B, H, S, D = 4, 32, 4096, 128
    gen = torch.Generator(device="cuda").manual_seed(1234)
    q, k, v = (torch.randn(B, H, S, D, device="cuda", dtype=torch.bfloat16, generator=gen) for _ in range(3))
    out = aule.flash_attention(q, k, v, causal=True)
    torch.cuda.synchronize()
    assert torch.isfinite(out.float()).all()
    # (1) sampled rows against the fp64 oracle
    rng = np.random.RandomState(0)
    rows = _sample_rows(rng, B * H * S, 48)
    qn, kn, vn = (x.float().cpu().numpy() for x in (q, k, v))
    ref, _ = oracle_mod.fwd_rows_f64(qn, kn, vn, rows, True, None)
    got = out.float().cpu().numpy().reshape(-1, D)[rows]
    assert_close_rows(got, ref, rows % S + 1, "bf16", v.float().abs().max().item(), "C2 sampled rows")
    # (2) causal prefix invariance: rows < 1024 do not depend on later keys (bit-exact: same tiles)
    out_p = aule.flash_attention(q[:, :, :1024].contiguous(), k[:, :, :1024].contiguous(),
                                 v[:, :, :1024].contiguous(), causal=True)
    assert torch.equal(out_p, out[:, :, :1024])
    # (3) batch independence (bit-exact)
    out_b = aule.flash_attention(q[2:3], k[2:3], v[2:3], causal=True)
    assert torch.equal(out_b, out[2:3])
    # (4) linearity in V: O(V1 + V2) ~= O(V1) + O(V2)
    v2 = torch.randn_like(v[:1])
    o1 = aule.flash_attention(q[:1], k[:1], v[:1], causal=True).float()
    o2 = aule.flash_attention(q[:1], k[:1], v2, causal=True).float()
    o12 = aule.flash_attention(q[:1], k[:1], (v[:1].float() + v2.float()).to(torch.bfloat16), causal=True).float()
    assert (o12 - (o1 + o2)).abs().max().item() < 0.05


def test_config4_shard_sampled_rows_and_properties(torch_cuda, oracle_mod):
    """BASELINE config #4, one GPU's shard of the batch-sharded 8-GPU problem: B=8 H=32 S=8192 D=128 bf16 causal fwd
    (B=64 over 8 ranks; the ranks are independent, aule/dist.py).  Sampled rows vs the fp64 judge, LSE included, and
    the size-independent properties."""
    import aule
    from aule import _torch as at
    torch = torch_cuda
    B, H, S, D = 8, 32, 8192, 128
    gen = torch.Generator(device="cuda").manual_seed(4321)
    q, k, v = (torch.randn(B, H, S, D, device="cuda", dtype=torch.bfloat16, generator=gen) for _ in range(3))
    out, lse = at.fwd_raw(q, k, v, True, 1.0 / math.sqrt(D))
    torch.cuda.synchronize()
    assert torch.isfinite(out.float()).all() and torch.isfinite(lse).all()
    rng = np.random.RandomState(5)
    rows = _sample_rows(rng, B * H * S, 40)
    rows = np.unique(np.concatenate([rows, [S - 1, (B * H - 1) * S, 255, 256, 8191 - 255]]))   # block edges, first / last head
    ref, ref_lse = oracle_mod.fwd_rows_f64(q.float().cpu().numpy(), k.float().cpu().numpy(), v.float().cpu().numpy(),
                                           rows, True, None)
    got = out.float().cpu().numpy().reshape(-1, D)[rows]
    assert_close_rows(got, ref, rows % S + 1, "bf16", v.float().abs().max().item(), "C4 shard sampled rows")
    assert_close(lse.cpu().numpy().reshape(-1)[rows], ref_lse, LSE_TOL["bf16"], 1e-5, "C4 shard LSE")
    print("C4 shard achieved: out max|err| %.3e, lse max|err| %.3e"
          % (np.abs(got - ref).max(), np.abs(lse.cpu().numpy().reshape(-1)[rows] - ref_lse).max()))
    # causal prefix invariance (bit-exact: the same tiles in the same order) and batch independence
    out_p = aule.flash_attention(q[:, :, :2048].contiguous(), k[:, :, :2048].contiguous(), v[:, :, :2048].contiguous(), causal=True)
    assert torch.equal(out_p, out[:, :, :2048])
    out_b = aule.flash_attention(q[5:6], k[5:6], v[5:6], causal=True)
    assert torch.equal(out_b, out[5:6])


def test_device_mismatch_is_an_error_not_a_fault(torch_cuda):
    """A tensor on the wrong device must raise in Python instead of handing the kernel a foreign pointer."""
    from aule import _torch as at
    torch = torch_cuda
    q = torch.randn(1, 2, 64, 64, device="cuda", dtype=torch.float16)
    kc = torch.randn(1, 2, 64, 64, dtype=torch.float16)   # CPU
    with pytest.raises(ValueError, match="must live on"):
        at.fwd_raw(q, kc, q, True, 0.125)


def test_config5_mqa_fp16_long_noncausal_sampled_rows(torch_cuda, oracle_mod):
    """BASELINE config #5: MQA 32q/1kv S=16384 D=64 fp16 non-causal (B=1)."""
    import aule
    torch = torch_cuda
    B, Hq, Hkv, S, D = 1, 32, 1, 16384, 64
    gen = torch.Generator(device="cuda").manual_seed(77)
    q = torch.randn(B, Hq, S, D, device="cuda", dtype=torch.float16, generator=gen)
    k = torch.randn(B, Hkv, S, D, device="cuda", dtype=torch.float16, generator=gen)
    v = torch.randn(B, Hkv, S, D, device="cuda", dtype=torch.float16, generator=gen)
    out = aule.flash_attention(q, k, v, causal=False)
    rng = np.random.RandomState(1)
    rows = _sample_rows(rng, B * Hq * S, 32)
    ref, _ = oracle_mod.fwd_rows_f64(q.float().cpu().numpy(), k.float().cpu().numpy(), v.float().cpu().numpy(),
                                     rows, False, None)
    got = out.float().cpu().numpy().reshape(-1, D)[rows]
    assert_close_rows(got, ref, np.full(len(rows), S), "fp16", v.float().abs().max().item(), "C5 sampled rows")
    # decode-like cross attention (Sq = 1 and 64) against the same keys
    for sq in (1, 64):
        o = aule.flash_attention(q[:, :, :sq].contiguous(), k, v, causal=False)
        r, _ = oracle_mod.fwd_f64(q[:, :, :sq].float().cpu().numpy(), k.float().cpu().numpy(),
                                  v.float().cpu().numpy(), False, None)
        assert_close(o.float().cpu().numpy(), r, *fwd_tol("fp16", v.float().abs().max().item()), f"C5 Sq={sq}")


def test_sdpa_shim_matches_torch_math(torch_cuda):
    """install(): F.scaled_dot_product_attention runs the HIP kernels (causal, GQA, custom scale) and
    agrees with PyTorch's own SDPA; unsupported arguments fall back."""
    import aule
    import torch.nn.functional as F
    torch = torch_cuda
    gen = torch.Generator(device="cuda").manual_seed(5)
    q = torch.randn(2, 8, 200, 64, device="cuda", dtype=torch.float16, generator=gen)
    k = torch.randn(2, 2, 200, 64, device="cuda", dtype=torch.float16, generator=gen)
    v = torch.randn(2, 2, 200, 64, device="cuda", dtype=torch.float16, generator=gen)
    want = F.scaled_dot_product_attention(q.float(), k.float().repeat_interleave(4, 1), v.float().repeat_interleave(4, 1),
                                          is_causal=True, scale=0.2)
    aule.install()
    try:
        got = F.scaled_dot_product_attention(q, k, v, is_causal=True, scale=0.2, enable_gqa=True)
        mask = torch.ones(200, 200, dtype=torch.bool, device="cuda").tril()
        fb = F.scaled_dot_product_attention(q[:, :2], k, v, attn_mask=mask)     # falls back to torch
    finally:
        aule.uninstall()
    assert got.dtype == torch.float16
    assert_close(got.float().cpu().numpy(), want.cpu().numpy(), *fwd_tol("fp16", v.float().abs().max().item()), "sdpa")
    assert tuple(fb.shape) == (2, 2, 200, 64)
