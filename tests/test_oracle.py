"""CPU tests that PIN the oracle (no GPU): known-answer tests the reference's own tests
hold, and the golden vectors generated from the reference's Python (tests/golden/)."""
import numpy as np
import pytest

from conftest import golden_files, load_golden


# ---- known-answer tests from the reference ------------------------------------------
def test_kat1_constant_qk_small(oracle_mod):
    """src/attention_ref.zig:250-298: Q=K=0.5, V=[[1,2,3,4],[5,6,7,8]] -> every row [3,4,5,6]."""
    q = np.full((1, 1, 2, 4), 0.5, np.float32)
    v = np.array([[1, 2, 3, 4], [5, 6, 7, 8]], np.float32).reshape(1, 1, 2, 4)
    out = oracle_mod.ref_forward(q, q, v)
    np.testing.assert_allclose(out.reshape(2, 4), [[3, 4, 5, 6], [3, 4, 5, 6]], atol=1e-3)
    out64, _ = oracle_mod.fwd_f64(q, q, v, causal=False)
    np.testing.assert_allclose(out64.reshape(2, 4), [[3, 4, 5, 6], [3, 4, 5, 6]], atol=1e-6)


def test_kat2_uniform_weights_mean_of_v(oracle_mod):
    """tests/test_attention.zig:158-219: Q=K=0.5, S=4, D=8, V[i*8+d]=i*8+d -> column means 12..19."""
    q = np.full((1, 1, 4, 8), 0.5, np.float32)
    v = np.arange(32, dtype=np.float32).reshape(1, 1, 4, 8)
    for out in (oracle_mod.ref_forward(q, q, v), oracle_mod.backend_cpu_attention(q, q, v),
                oracle_mod.cpu_attention(q, q, v, causal=False)):
        np.testing.assert_allclose(out.reshape(4, 8), np.tile(np.arange(12, 20), (4, 1)), atol=0.01)


def test_kat3_one_hot_selects_v(oracle_mod):
    """tests/test_attention.zig:221-270: Q,K = 10*one-hot on the diagonal, V[i]=0.1*i -> |O-V| < 0.1."""
    S = D = 8
    q = (10.0 * np.eye(S, D, dtype=np.float32)).reshape(1, 1, S, D)
    v = (0.1 * np.arange(S, dtype=np.float32))[:, None].repeat(D, 1).reshape(1, 1, S, D)
    out = oracle_mod.ref_forward(q, q, v)
    assert np.abs(out - v).max() < 0.1


def test_kat4_batch_independence(oracle_mod):
    """tests/test_attention.zig:327-384: batches do not interact (< 1e-5)."""
    rng = np.random.RandomState(3)
    q, k, v = (rng.uniform(-0.5, 0.5, (2, 2, 16, 16)).astype(np.float32) for _ in range(3))
    both = oracle_mod.ref_forward(q, k, v)
    for b in range(2):
        single = oracle_mod.ref_forward(q[b:b + 1], k[b:b + 1], v[b:b + 1])
        assert np.abs(both[b:b + 1] - single).max() < 1e-5


def test_finite_on_pm5_inputs(oracle_mod):
    """tests/test_attention.zig:272-325: +-5 range stays finite."""
    rng = np.random.RandomState(4)
    q, k, v = (rng.uniform(-5, 5, (1, 2, 32, 32)).astype(np.float32) for _ in range(3))
    for causal in (False, True):
        assert np.isfinite(oracle_mod.ref_forward(q, k, v, causal)).all()


# ---- the restatements agree with each other -------------------------------------------
@pytest.mark.parametrize("causal", [False, True])
def test_zig_order_vs_f64_vs_numpy(oracle_mod, small_qkv, causal):
    q, k, v = small_qkv
    a = oracle_mod.ref_forward(q, k, v, causal)
    b, _ = oracle_mod.fwd_f64(q, k, v, causal)
    c = oracle_mod.cpu_attention(q, k, v, causal)
    # tolerance rule of tests/test_attention.zig:69-76: max_abs < 1e-4 or max_rel < 1e-3
    assert np.abs(a - b).max() < 1e-4
    assert np.abs(c - b).max() < 1e-4
    if not causal:
        assert np.abs(oracle_mod.backend_cpu_attention(q, k, v) - b).max() < 1e-4


def test_c_oracle_matches_numpy_f64(oracle_mod):
    rng = np.random.RandomState(5)
    q = rng.randn(2, 6, 20, 16).astype(np.float32)
    k = rng.randn(2, 2, 28, 16).astype(np.float32)
    v = rng.randn(2, 2, 28, 16).astype(np.float32)
    do = rng.randn(2, 6, 20, 16).astype(np.float32)
    for causal in (False, True):
        o, l = oracle_mod.fwd_f64(q, k, v, causal, 0.37)
        o2, l2 = oracle_mod.np_fwd_f64(q, k, v, causal, 0.37)
        assert np.abs(o - o2).max() < 1e-6 and np.abs(l - l2).max() < 1e-6
        for g, g2 in zip(oracle_mod.bwd_f64(q, k, v, do, causal, 0.37),
                         oracle_mod.np_bwd_f64(q, k, v, do, causal, 0.37)):
            assert np.abs(g - g2).max() < 1e-5
        rows = np.array([0, 7, 19, 2 * 6 * 20 - 1, 123], dtype=np.int64)
        orows, lrows = oracle_mod.fwd_rows_f64(q, k, v, rows, causal, 0.37)
        assert np.abs(orows - o.reshape(-1, 16)[rows]).max() < 1e-7
        assert np.abs(lrows - l.reshape(-1)[rows]).max() < 1e-7


def test_bf16_quantiser_matches_torch(oracle_mod):
    import torch
    rng = np.random.RandomState(6)
    a = (rng.randn(4096) * np.exp(rng.randn(4096) * 4)).astype(np.float32)
    want = torch.from_numpy(a).to(torch.bfloat16).float().numpy()
    assert np.array_equal(oracle_mod.quantize_bf16(a), want)


# ---- golden vectors generated from the reference ---------------------------------------
@pytest.mark.parametrize("path", golden_files("np_"), ids=lambda p: p.split("/")[-1][:-4])
def test_numpy_fallback_goldens(oracle_mod, path):
    """python/aule/__init__.py:247-271 output vs our numpy restatement (same arithmetic,
    rtol=atol=1e-4 is the reference's own bar, python/tests/test_cpu.py:29; we hold 1e-6)
    and vs the fp64 judge (1e-5, BASELINE.json fp32 bar)."""
    g = load_golden(path)
    out = oracle_mod.cpu_attention(g["q"], g["k"], g["v"], g["causal"])
    np.testing.assert_allclose(out, g["out"], rtol=1e-6, atol=1e-6)
    B, Hq, Hkv, Sq, Sk, D = [int(x) for x in g["shape"]]
    o64, _ = oracle_mod.fwd_f64(g["q"], g["k"], g["v"], g["causal"])
    np.testing.assert_allclose(o64, g["out"], rtol=1e-5, atol=1e-5)
    if Sq == Sk:
        np.testing.assert_allclose(oracle_mod.ref_forward(g["q"], g["k"], g["v"], g["causal"]), g["out"],
                                   rtol=1e-4, atol=1e-5)


FWD_GOLD_TOL = {"fp32": 1e-5, "fp16": 2e-3, "bf16": 2e-2}   # golden includes the storage rounding of O


@pytest.mark.parametrize("path", golden_files("tr_") + golden_files("amd_"), ids=lambda p: p.split("/")[-1][:-4])
def test_triton_goldens_forward_and_lse(oracle_mod, path):
    """Reference Triton kernels (interpreted) vs the fp64 judge on the captured inputs:
    O, and LSE = m + ln(l) over scaled scores (triton_flash_amd.py:237)."""
    g = load_golden(path)
    o, lse = oracle_mod.fwd_f64(g["q"], g["k"], g["v"], g["causal"], g["scale"])
    tol = FWD_GOLD_TOL[g["dtype"]]
    np.testing.assert_allclose(o, g["out"], rtol=tol, atol=tol)
    np.testing.assert_allclose(lse, g["lse"], rtol=1e-5, atol=2e-5 if g["dtype"] == "fp32" else 2e-3)


@pytest.mark.parametrize("path", [p for p in golden_files("tr_") if "dq" in np.load(p).files],
                         ids=lambda p: p.split("/")[-1][:-4])
def test_triton_goldens_backward(oracle_mod, path):
    """dQ, dK, dV of the reference (triton_flash.py:478-526) vs the fp64 judge."""
    g = load_golden(path)
    dq, dk, dv = oracle_mod.bwd_f64(g["q"], g["k"], g["v"], g["dout"], g["causal"], g["scale"])
    tol = 2e-5 if g["dtype"] == "fp32" else 2e-2
    np.testing.assert_allclose(dq, g["dq"], rtol=tol, atol=tol)
    np.testing.assert_allclose(dk, g["dk"], rtol=tol, atol=tol)
    np.testing.assert_allclose(dv, g["dv"], rtol=tol, atol=tol)


@pytest.mark.parametrize("path", golden_files("win_"), ids=lambda p: p.split("/")[-1][:-4])
def test_window_goldens_pin_the_windowed_oracle(oracle_mod, path):
    """Sliding window (row N1): the reference's ROCm kernel with window_size > 0 (triton_flash_amd.py:179-183,
    interpreted; fixtures from gen_golden.py window) vs the C judge and its NumPy twin."""
    g = load_golden(path)
    W = int(np.load(path)["window"])
    o, lse = oracle_mod.fwd_f64(g["q"], g["k"], g["v"], g["causal"], g["scale"], W)
    np.testing.assert_allclose(o, g["out"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(lse, g["lse"], rtol=1e-5, atol=2e-5)
    o2, lse2 = oracle_mod.np_fwd_f64(g["q"], g["k"], g["v"], g["causal"], g["scale"], W)
    np.testing.assert_allclose(o2, g["out"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(lse2, g["lse"], rtol=1e-5, atol=2e-5)


def test_windowed_backward_judges_agree(oracle_mod):
    """No reference golden exists for the windowed gradients (the reference's backward drops the window):
    the C judge and the NumPy judge are two independent restatements of Appendix B with the window added."""
    rng = np.random.RandomState(4)
    q, do = rng.randn(1, 4, 70, 16).astype(np.float32), rng.randn(1, 4, 70, 16).astype(np.float32)
    k, v = rng.randn(1, 2, 90, 16).astype(np.float32), rng.randn(1, 2, 90, 16).astype(np.float32)
    for causal in (True, False):
        for W in (1, 7, 33, 200):
            a = oracle_mod.bwd_f64(q, k, v, do, causal, None, W)
            b = oracle_mod.np_bwd_f64(q, k, v, do, causal, None, W)
            for x, y in zip(a, b):
                np.testing.assert_allclose(x, y, rtol=1e-5, atol=1e-6)
    # a window that masks nothing is the unwindowed judge
    for x, y in zip(oracle_mod.bwd_f64(q, k, v, do, True, None, 10 ** 6), oracle_mod.bwd_f64(q, k, v, do, True)):
        np.testing.assert_array_equal(x, y)


def test_per_head_backward_judge_is_pinned_on_the_c_judge(oracle_mod):
    """oracle.bwd_head_f64 (round 6: one KV head + its query group at full S, row-blocked BLAS fp64 -- what the C2 / C4-shard gradient
    tests use) against the scalar C judge of the whole problem: GQA groups, ragged sizes that straddle its row blocks, every causal
    mode, windows, a custom scale, and a block size that does not divide Sq."""
    rng = np.random.RandomState(6)
    for (B, Hq, Hkv, Sq, Sk, D, causal, scale, W, blk) in (
            (2, 4, 2, 70, 90, 16, True, None, -1, 32), (1, 6, 3, 97, 97, 32, False, 0.37, -1, 40), (1, 4, 1, 50, 130, 16, 2, None, -1, 16),
            (1, 2, 2, 130, 130, 16, True, None, 33, 64), (1, 4, 2, 64, 200, 8, 2, 0.2, 7, 24), (1, 2, 1, 150, 60, 16, True, None, 5, 512),
            (1, 2, 2, 90, 90, 16, False, None, 20, 32)):
        q, do = (rng.randn(B, Hq, Sq, D).astype(np.float32) for _ in range(2))
        k, v = (rng.randn(B, Hkv, Sk, D).astype(np.float32) for _ in range(2))
        rq, rk, rv = oracle_mod.bwd_f64(q, k, v, do, causal, scale, W)
        g = Hq // Hkv
        for b in range(B):
            for hk in range(Hkv):
                dq, dk, dv = oracle_mod.bwd_head_f64(q, k, v, do, (b, hk), causal, scale, W, block=blk)
                np.testing.assert_allclose(dq, rq[b, hk * g:(hk + 1) * g], rtol=1e-5, atol=1e-6)
                np.testing.assert_allclose(dk, rk[b, hk], rtol=1e-5, atol=1e-6)
                np.testing.assert_allclose(dv, rv[b, hk], rtol=1e-5, atol=1e-6)
    # the sliced calling form is the same function
    a = oracle_mod.bwd_head_f64(q[0, 0:1], k[0, 0], v[0, 0], do[0, 0:1], None, False, None, 20)
    for x, y in zip(a, oracle_mod.bwd_head_f64(q, k, v, do, (0, 0), False, None, 20)):
        np.testing.assert_array_equal(x, y)


@pytest.mark.parametrize("path", [p for p in golden_files("tr_") if "dq" in np.load(p).files][:4], ids=lambda p: p.split("/")[-1][:-4])
def test_per_head_backward_judge_vs_reference_gradients(oracle_mod, path):
    """... and on the gradients the reference's own kernels recorded (tests/golden/tr_*: Triton under TRITON_INTERPRET=1), head by head,
    at the tolerance the whole-problem judge is held to against them."""
    g = load_golden(path)
    q, k, v, do = g["q"], g["k"], g["v"], g["dout"]
    grp = q.shape[1] // k.shape[1]
    tol = 2e-2 if g["dtype"] != "fp32" else 2e-5
    for b in range(q.shape[0]):
        for hk in range(k.shape[1]):
            dq, dk, dv = oracle_mod.bwd_head_f64(q, k, v, do, (b, hk), g["causal"], g["scale"])
            np.testing.assert_allclose(dq, g["dq"][b, hk * grp:(hk + 1) * grp], rtol=tol, atol=tol)
            np.testing.assert_allclose(dk, g["dk"][b, hk], rtol=tol, atol=tol)
            np.testing.assert_allclose(dv, g["dv"][b, hk], rtol=tol, atol=tol)


def test_bottom_right_oracle_pinned_on_torch_lower_right_bias(oracle_mod):
    """The bottom-right alignment is not in the reference; the oracle's causal=2 mode is pinned on
    PyTorch's own definition of it (torch.nn.attention.bias.causal_lower_right) through fp64 SDPA with autograd,
    and on the two judges (C and NumPy) agreeing, with and without a window."""
    import torch
    from torch.nn.attention.bias import causal_lower_right
    rng = np.random.RandomState(8)
    Sq, Sk, D = 37, 90, 16
    q, do = rng.randn(1, 4, Sq, D).astype(np.float32), rng.randn(1, 4, Sq, D).astype(np.float32)
    k, v = rng.randn(1, 2, Sk, D).astype(np.float32), rng.randn(1, 2, Sk, D).astype(np.float32)
    tq, tk, tv = (torch.from_numpy(x).double().requires_grad_(True) for x in (q, k, v))
    bias = causal_lower_right(Sq, Sk)._materialize()          # bool [Sq, Sk], True = visible
    assert bool(bias[0, Sk - Sq]) and not bool(bias[0, Sk - Sq + 1]) and bool(bias[-1].all())
    ref = torch.nn.functional.scaled_dot_product_attention(
        tq, tk.repeat_interleave(2, 1), tv.repeat_interleave(2, 1), attn_mask=bias)
    ref.backward(torch.from_numpy(do).double())
    out, lse = oracle_mod.fwd_f64(q, k, v, "bottom-right")
    np.testing.assert_allclose(out, ref.detach().numpy(), rtol=1e-5, atol=1e-6)
    dq, dk, dv = oracle_mod.bwd_f64(q, k, v, do, "bottom-right")
    np.testing.assert_allclose(dq, tq.grad.numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(dk, tk.grad.numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(dv, tv.grad.numpy(), rtol=1e-5, atol=1e-6)
    for W in (-1, 7, 33):
        a = oracle_mod.fwd_f64(q, k, v, 2, None, W)
        b = oracle_mod.np_fwd_f64(q, k, v, 2, None, W)
        np.testing.assert_allclose(a[0], b[0], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(a[1], b[1], rtol=1e-5, atol=1e-5)
        for x, y in zip(oracle_mod.bwd_f64(q, k, v, do, 2, None, W), oracle_mod.np_bwd_f64(q, k, v, do, 2, None, W)):
            np.testing.assert_allclose(x, y, rtol=1e-5, atol=1e-6)
    # Sq == Sk: the two alignments are the same mask, bit for bit
    qs = rng.randn(1, 4, Sk, D).astype(np.float32)
    for x, y in zip(oracle_mod.fwd_f64(qs, k, v, 2), oracle_mod.fwd_f64(qs, k, v, True)):
        np.testing.assert_array_equal(x, y)
    # the sampled-rows judge follows the same rule
    rows = np.array([0, 5, Sq - 1, Sq + 3, 4 * Sq - 1], dtype=np.int64)
    o_r, l_r = oracle_mod.fwd_rows_f64(q, k, v, rows, 2)
    np.testing.assert_allclose(o_r, out.reshape(-1, D)[rows], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(l_r, lse.reshape(-1)[rows], rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("path", golden_files("rope_"), ids=lambda p: p.split("/")[-1][:-4])
def test_rope_goldens_pin_the_rope_oracle(oracle_mod, path):
    """tests/golden/rope_*: the reference's tables (triton_flash.py:644-677), its half-split rotation (:680-703) and
    its FA-2 kernel over the rotated Q, K (what its self-test :788-806 expects of the fused path)."""
    g = np.load(path)
    D = g["q"].shape[-1]
    c, s = oracle_mod.rope_tables(g["cos"].shape[0], D)
    np.testing.assert_allclose(c, g["cos"], rtol=0, atol=1e-5)
    np.testing.assert_allclose(s, g["sin"], rtol=0, atol=1e-5)
    qr = oracle_mod.rope_f64(g["q"], g["cos"], g["sin"], "half")
    kr = oracle_mod.rope_f64(g["k"], g["cos"], g["sin"], "half")
    np.testing.assert_allclose(qr, g["q_rot"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(kr, g["k_rot"], rtol=1e-6, atol=1e-6)
    out, _ = oracle_mod.fwd_f64(qr, kr, g["v"], bool(g["causal"]))
    np.testing.assert_allclose(out, g["out"], rtol=1e-5, atol=1e-5)


def test_rope_oracle_layouts_and_inverse(oracle_mod):
    """Interleaved pairs restated from the reference's own test oracle (tests/test_rope_unit.py:76-86: view
    [.., D/2, 2], x1' = x1 c - x2 s, x2' = x1 s + x2 c), the two layouts agreeing up to the pair permutation,
    and the inverse being the transpose."""
    rng = np.random.RandomState(12)
    x = rng.randn(2, 3, 11, 16).astype(np.float32)
    cos, sin = oracle_mod.rope_tables(20, 16)
    got = oracle_mod.rope_f64(x, cos, sin, "interleaved", pos_offset=4)
    xr = x.astype(np.float64).reshape(2, 3, 11, 8, 2)
    c, s = cos[4:15].astype(np.float64), sin[4:15].astype(np.float64)
    want = np.stack([xr[..., 0] * c - xr[..., 1] * s, xr[..., 0] * s + xr[..., 1] * c], axis=-1).reshape(x.shape)
    np.testing.assert_allclose(got, want, rtol=1e-6, atol=1e-6)
    perm = np.concatenate([np.arange(0, 16, 2), np.arange(1, 16, 2)])         # interleaved -> half-split order
    half = oracle_mod.rope_f64(x[..., perm], cos, sin, "half", pos_offset=4)
    np.testing.assert_allclose(half, got[..., perm], rtol=1e-6, atol=1e-6)
    for layout in ("half", "interleaved"):
        y = oracle_mod.rope_f64(x, cos, sin, layout)
        np.testing.assert_allclose(oracle_mod.rope_f64(y, cos, sin, layout, inverse=True), x, rtol=1e-5, atol=1e-6)
        # <R a, b> = <a, R^T b>
        b = rng.randn(*x.shape).astype(np.float32)
        lhs = (y.astype(np.float64) * b).sum()
        rhs = (x.astype(np.float64) * oracle_mod.rope_f64(b, cos, sin, layout, inverse=True)).sum()
        assert abs(lhs - rhs) < 1e-4 * max(1.0, abs(lhs))


@pytest.mark.parametrize("path", golden_files("paged_"), ids=lambda p: p.split("/")[-1][:-4])
def test_paged_goldens_pin_the_paged_oracle(oracle_mod, path):
    """Paged-KV decode (row N2): the reference's flash_attention_paged_amd (interpreted, fp16 I/O) vs the judge."""
    z = np.load(path)
    o = oracle_mod.paged_decode_f64(z["q"], z["k_cache"], z["v_cache"], z["block_tables"], z["context_lens"],
                                    None, int(z["window"]))
    np.testing.assert_allclose(o, z["out"], rtol=2e-3, atol=2e-3)   # golden output is rounded to fp16


def test_c1_fixture_is_reference_config():
    g = load_golden([p for p in golden_files("np_") if "c1_" in p][0])
    assert [int(x) for x in g["shape"]] == [1, 8, 8, 256, 256, 64] and g["causal"] and g["dtype"] == "fp32"
    import hashlib
    h = hashlib.sha256()
    for a in (g["q"], g["k"], g["v"]):
        h.update(np.ascontiguousarray(a).tobytes())
    assert h.hexdigest() == g["input_sha256"], "regenerated inputs differ from the ones the golden was made from"
