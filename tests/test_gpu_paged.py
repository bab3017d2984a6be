"""Paged-KV decode on the GPU (csrc/fa_fwd_splitkv_gfx950.hip PAGED variant behind
aule.flash_attention_paged_amd / aule_attention_paged_decode_ex): golden vectors recorded from the reference's
flash_attention_paged_amd, the fp64 judge on larger seeded problems (GQA, MQA, ragged context lengths including 0
and 1, shuffled block tables, block sizes 8..128, window), and agreement with the contiguous kernel."""
import numpy as np
import pytest

from conftest import golden_files
from util import assert_close, fwd_tol, quantize, torch_dtype

pytestmark = pytest.mark.gpu


def _run(torch, q, kc, vc, bt, cl, dtype, scale=None, window=-1):
    import aule
    dt = torch_dtype(dtype)
    out = aule.flash_attention_paged_amd(torch.from_numpy(q).to("cuda", dt), torch.from_numpy(kc).to("cuda", dt),
                                         torch.from_numpy(vc).to("cuda", dt), torch.from_numpy(bt).cuda(),
                                         torch.from_numpy(cl).cuda(), scale=scale, window_size=window)
    return out.float().cpu().numpy()


@pytest.mark.parametrize("path", golden_files("paged_"), ids=lambda p: p.split("/")[-1][:-4])
def test_paged_goldens(path):
    import torch
    z = np.load(path)
    out = _run(torch, z["q"], z["k_cache"], z["v_cache"], z["block_tables"], z["context_lens"], str(z["dtype"]),
               None, int(z["window"]))
    atol, rtol = fwd_tol(str(z["dtype"]), np.abs(z["v_cache"]).max(), sides=2)
    assert_close(out, z["out"], atol, rtol, "paged golden")


CASES = [  # dtype, B, Hq, Hkv, D, block_size, context lens, window
    ("bf16", 4, 32, 8, 128, 16, [1000, 37, 4096, 1], -1),
    ("fp16", 2, 32, 1, 64, 128, [5000, 129], -1),
    ("bf16", 3, 8, 8, 128, 8, [0, 77, 300], -1),          # a sequence with no key: zeros
    ("fp16", 2, 16, 4, 32, 32, [2048, 2047], 256),        # sliding window over the last 256 positions
    ("bf16", 1, 64, 8, 128, 64, [20000], -1),
    # block sizes that are NOT powers of two take the general address path (divide / modulo per key row; the
    # power-of-two sizes above take the shift / mask fast path), and 32-key tiles straddle their block boundaries
    ("bf16", 3, 16, 4, 128, 48, [1000, 47, 4000], -1),
    ("fp16", 2, 32, 8, 64, 24, [3001, 25], -1),
    ("fp16", 2, 8, 2, 32, 100, [2500, 99], 300),
    ("bf16", 2, 8, 8, 128, 1, [700, 3], -1),             # one token per block
    ("bf16", 2, 32, 4, 64, 33, [5000, 1], 64),
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: f"{c[0]}-B{c[1]}-H{c[2]}kv{c[3]}-D{c[4]}-bs{c[5]}-w{c[7]}")
def test_paged_vs_oracle(case, oracle_mod):
    import torch
    dtype, B, Hq, Hkv, D, bs, lens, window = case
    rng = np.random.RandomState(31)
    nblk = [(n + bs - 1) // bs for n in lens]
    num_blocks = sum(nblk) + 3
    q = quantize(rng.randn(B, Hq, D).astype(np.float32), dtype)
    kc = quantize(rng.randn(num_blocks, bs, Hkv, D).astype(np.float32), dtype)
    vc = quantize(rng.randn(num_blocks, bs, Hkv, D).astype(np.float32), dtype)
    bt = np.full((B, max(max(nblk), 1) + 2), 0, dtype=np.int32)    # unused columns point at block 0
    perm = rng.permutation(num_blocks)
    used = 0
    for b in range(B):
        bt[b, :nblk[b]] = perm[used:used + nblk[b]]
        used += nblk[b]
    cl = np.array(lens, dtype=np.int32)
    out = _run(torch, q, kc, vc, bt, cl, dtype, None, window)
    ref = oracle_mod.paged_decode_f64(q, kc, vc, bt, cl, None, window)
    atol, rtol = fwd_tol(dtype, np.abs(vc).max())
    assert_close(out, ref, atol, rtol, "paged")


def _paged_view(k, v, bs):
    """cache [num_blocks, bs, Hkv, D] holding sequence b in blocks b*nb .. (b+1)*nb-1, identity block table"""
    import torch
    B, Hkv, n, D = k.shape
    nb = n // bs
    kc = k.permute(0, 2, 1, 3).reshape(B * nb, bs, Hkv, D).contiguous()
    vc = v.permute(0, 2, 1, 3).reshape(B * nb, bs, Hkv, D).contiguous()
    bt = torch.arange(B * nb, device="cuda", dtype=torch.int32).view(B, nb)
    cl = torch.full((B,), n, device="cuda", dtype=torch.int32)
    return kc, vc, bt, cl


def _dense_route(B, Hq, Hkv, Sk, D):
    import ctypes
    from aule import _capi
    d = _capi.AttnDesc()
    d.struct_size = ctypes.sizeof(_capi.AttnDesc)
    d.dtype = 2
    d.batch, d.heads_q, d.heads_kv, d.seq_q, d.seq_k, d.head_dim = B, Hq, Hkv, 1, Sk, D
    d.causal, d.window_size = 0, -1
    return _capi.get_lib().aule_hip_debug_forward_route(ctypes.byref(d))


def test_paged_context_len_beyond_the_block_table_is_clamped():
    """A context length larger than max_blocks * block_size (a stale scheduler value) is clamped on the device to what
    the table can address -- same result as the clamped length, no out-of-bounds table or cache read -- and the block
    table / lengths may arrive as CPU tensors (they are moved to the query's device, not read through a host pointer)."""
    import torch
    import aule
    torch.manual_seed(9)
    B, Hq, Hkv, D, bs, nb = 2, 8, 2, 128, 16, 6
    q = torch.randn(B, Hq, D, device="cuda", dtype=torch.float16)
    kc = torch.randn(B * nb, bs, Hkv, D, device="cuda", dtype=torch.float16)
    vc = torch.randn_like(kc)
    bt = torch.arange(B * nb, dtype=torch.int32).reshape(B, nb)            # CPU on purpose
    full = torch.tensor([nb * bs, nb * bs], dtype=torch.int32)
    over = torch.tensor([nb * bs + 1000, 2 ** 30], dtype=torch.int32)
    ref = aule.flash_attention_paged_amd(q, kc, vc, bt, full)
    got = aule.flash_attention_paged_amd(q, kc, vc, bt, over)
    torch.cuda.synchronize()
    assert torch.equal(got, ref)


def test_paged_equals_contiguous_decode():
    """Identity block table + one context length = decode on contiguous K/V [B,Hkv,Sk,D].

    Small problem: the contiguous side runs the tiled kernel with packed rows + KV splits (route 5), the paged side the
    wave-per-chunk kernel, so the two differ by summation order only -- asserted to be within bf16 rounding of each
    other (both are checked against the fp64 oracle elsewhere)."""
    import torch
    import aule
    torch.manual_seed(5)
    B, Hq, Hkv, D, bs, n = 2, 16, 4, 128, 32, 2048
    k = torch.randn(B, Hkv, n, D, device="cuda", dtype=torch.bfloat16)
    v = torch.randn(B, Hkv, n, D, device="cuda", dtype=torch.bfloat16)
    q = torch.randn(B, Hq, 1, D, device="cuda", dtype=torch.bfloat16)
    assert _dense_route(B, Hq, Hkv, n, D) == 5
    dense = aule.flash_attention(q, k, v, causal=False)                      # [B,Hq,1,D]
    kc, vc, bt, cl = _paged_view(k, v, bs)
    paged = aule.flash_attention_paged_amd(q, kc, vc, bt, cl)
    assert paged.shape == (B, Hq, D)
    diff = float((paged.float() - dense.squeeze(2).float()).abs().max())
    # two roundings of the same value to bf16 differ by at most one ulp, and ulp(x) <= 2^-7 |x|
    assert diff <= 2.0 ** -7 * float(dense.float().abs().max()), diff


def test_paged_is_bit_identical_to_contiguous_on_the_wave_kernel():
    """Where the rule sends contiguous decode to the wave-per-chunk kernel too (>= 32 units, <= 16 packed rows,
    K+V >= 100 MB), paged and contiguous are the same arithmetic in the same order: bit for bit."""
    import torch
    import aule
    torch.manual_seed(6)
    B, Hq, Hkv, D, bs, n = 8, 32, 8, 128, 32, 8192
    k = torch.randn(B, Hkv, n, D, device="cuda", dtype=torch.bfloat16)
    v = torch.randn(B, Hkv, n, D, device="cuda", dtype=torch.bfloat16)
    q = torch.randn(B, Hq, 1, D, device="cuda", dtype=torch.bfloat16)
    assert _dense_route(B, Hq, Hkv, n, D) == 4
    dense = aule.flash_attention(q, k, v, causal=False)
    kc, vc, bt, cl = _paged_view(k, v, bs)
    paged = aule.flash_attention_paged_amd(q, kc, vc, bt, cl)
    assert torch.equal(paged, dense.squeeze(2))
