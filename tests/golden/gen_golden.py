#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE implementation.

Runs ONLY in the build container, where the reference is mounted read-only at
/root/reference.  It imports the reference's Python package (NumPy fallback
`aule._cpu_attention`, and the Triton FlashAttention-2 kernels executed on CPU
tensors under TRITON_INTERPRET=1) and records inputs + outputs as small data
fixtures.  Nothing of the reference's source travels: the fixtures are arrays.

    TRITON_INTERPRET=1 PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_golden.py

Input convention (mirrors python/tests/conftest.py:10-15 of the reference):
`np.random.seed(seed)` then `randn` for q, k, v (then dO) in that order, float32.
Low-precision cases quantise those arrays with torch (`.to(bfloat16/float16)`)
and store the quantised values as float32 so no RNG/rounding rule has to be
re-derived on the GPU box.
"""
import hashlib
import math
import os
import sys

REF = "/root/reference/python"
if not os.path.isdir(REF):
    sys.exit("reference not mounted at /root/reference -- golden vectors can only be regenerated "
             "in the build container")
os.environ.setdefault("TRITON_INTERPRET", "1")
os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True
sys.path.insert(0, REF)

import warnings

import numpy as np
import torch

import aule  # the reference package
from aule.triton_flash import FlashAttentionTritonFunc
import aule.triton_flash_amd as ref_amd
import triton

OUT = os.path.dirname(os.path.abspath(__file__))


def make_inputs(seed, B, Hq, Hkv, Sq, Sk, D, with_do=False):
    np.random.seed(seed)
    q = np.random.randn(B, Hq, Sq, D).astype(np.float32)
    k = np.random.randn(B, Hkv, Sk, D).astype(np.float32)
    v = np.random.randn(B, Hkv, Sk, D).astype(np.float32)
    if with_do:
        do = np.random.randn(B, Hq, Sq, D).astype(np.float32)
        return q, k, v, do
    return q, k, v


def sha(*arrs):
    h = hashlib.sha256()
    for a in arrs:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


TORCH_DT = {"fp32": torch.float32, "fp16": torch.float16, "bf16": torch.bfloat16}


def numpy_cases():
    """python/aule/__init__.py:247-271 via aule.flash_attention on NumPy input
    (the reference's CPU route: __init__.py:191-193,:244)."""
    cases = [
        # name, seed, B, H, Sq, Sk, D, causal, dtype, store_inputs
        ("c1_b1h8s256d64_causal", 42, 1, 8, 256, 256, 64, True, "fp32", False),   # BASELINE config #1
        ("b1h4s32d64_causal", 42, 1, 4, 32, 32, 64, True, "fp32", True),          # conftest small
        ("b1h4s32d64_full", 42, 1, 4, 32, 32, 64, False, "fp32", True),
        ("b4h8s64d64_causal", 42, 4, 8, 64, 64, 64, True, "fp32", False),         # conftest batched
        ("b2h4s32d32_causal", 7, 2, 4, 32, 32, 32, True, "fp32", True),
        ("b2h4s32d128_causal", 8, 2, 4, 32, 32, 128, True, "fp32", True),
        ("b1h2s1d64_causal", 9, 1, 2, 1, 1, 64, True, "fp32", True),
        ("b1h2s17d64_causal", 10, 1, 2, 17, 17, 64, True, "fp32", True),
        ("b1h2sq16sk32d64_causal", 11, 1, 2, 16, 32, 64, True, "fp32", True),     # cross, top-left mask
        ("b1h2sq16sk32d64_full", 11, 1, 2, 16, 32, 64, False, "fp32", True),
        ("b1h2sq48sk20d32_causal", 12, 1, 2, 48, 20, 32, True, "fp32", True),     # Sq > Sk
    ]
    for name, seed, B, H, Sq, Sk, D, causal, dt, store in cases:
        q, k, v = make_inputs(seed, B, H, H, Sq, Sk, D)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            out_direct = aule._cpu_attention(q, k, v, causal)
        rec = dict(kind="numpy_cpu", seed=seed, shape=np.array([B, H, H, Sq, Sk, D]), causal=causal,
                   dtype=dt, input_sha256=sha(q, k, v), out=out_direct.astype(np.float32))
        if store:
            rec.update(q=q, k=k, v=v)
        np.savez_compressed(os.path.join(OUT, f"np_{name}.npz"), **rec)
        print("numpy", name, out_direct.shape, float(np.abs(out_direct).max()))


def triton_cases():
    """python/aule/triton_flash.py:386-526 (generic FA-2 fwd+bwd) under the Triton
    interpreter: O, LSE (ctx.saved_tensors[4]), dQ, dK, dV."""
    cases = [
        # name, seed, B, Hq, Hkv, Sq, Sk, D, causal, scale, dtype, bwd
        ("mha_b2h4s64d64_causal", 42, 2, 4, 4, 64, 64, 64, True, None, "fp32", True),    # test_triton.py:41
        ("mha_b1h4s64d64_full", 43, 1, 4, 4, 64, 64, 64, False, None, "fp32", True),
        ("gqa_b1h12kv2s64d64_causal", 44, 1, 12, 2, 64, 64, 64, True, None, "fp32", True),  # test_triton.py:102-105
        ("mqa_b1h8kv1s64d64_causal", 45, 1, 8, 1, 64, 64, 64, True, None, "fp32", True),   # :119-122
        ("mha_b1h2s64d128_causal", 46, 1, 2, 2, 64, 64, 128, True, None, "fp32", True),    # :136
        ("gqa_b1h4kv2sq64sk96d32_causal", 47, 1, 4, 2, 64, 96, 32, True, None, "fp32", True),
        ("gqa_b1h4kv2sq64sk96d32_full", 47, 1, 4, 2, 64, 96, 32, False, None, "fp32", True),
        ("gqa_b1h4kv2s80d64_scale05", 48, 1, 4, 2, 80, 80, 64, True, 0.5, "fp32", True),
        ("mha_b1h2s33d64_causal", 49, 1, 2, 2, 33, 33, 64, True, None, "fp32", True),      # ragged
        ("gqa_b1h8kv2s128d128_bf16", 50, 1, 8, 2, 128, 128, 128, True, None, "bf16", True),
        ("gqa_b1h4kv1s128d128_bf16_full", 51, 1, 4, 1, 128, 128, 128, False, None, "bf16", True),
        ("mha_b1h4s128d64_fp16", 52, 1, 4, 4, 128, 128, 64, True, None, "fp16", True),     # test_triton.py:148
        ("mqa_b1h4kv1sq64sk160d64_fp16_full", 53, 1, 4, 1, 64, 160, 64, False, None, "fp16", True),
        ("gqa_b1h4kv2s192d128_bf16_causal", 54, 1, 4, 2, 192, 192, 128, True, None, "bf16", False),
    ]
    for name, seed, B, Hq, Hkv, Sq, Sk, D, causal, scale, dt, bwd in cases:
        q, k, v, do = make_inputs(seed, B, Hq, Hkv, Sq, Sk, D, with_do=True)
        tdt = TORCH_DT[dt]
        tq = torch.from_numpy(q).to(tdt).requires_grad_(True)
        tk = torch.from_numpy(k).to(tdt).requires_grad_(True)
        tv = torch.from_numpy(v).to(tdt).requires_grad_(True)
        tdo = torch.from_numpy(do).to(tdt)

        out = FlashAttentionTritonFunc.apply(tq, tk, tv, causal, scale, -1, None, None)
        lse = out.grad_fn.saved_tensors[4] if hasattr(out.grad_fn, "saved_tensors") else None
        rec = dict(kind="triton_generic", seed=seed, shape=np.array([B, Hq, Hkv, Sq, Sk, D]),
                   causal=causal, scale=(-1.0 if scale is None else float(scale)), dtype=dt,
                   q=tq.detach().float().numpy(), k=tk.detach().float().numpy(),
                   v=tv.detach().float().numpy(), dout=tdo.float().numpy(),
                   out=out.detach().float().numpy(), lse=lse.detach().float().numpy())
        if bwd:
            # The reference accumulates dQ/dK/dV with tl.atomic_add in the I/O dtype
            # (triton_flash.py:336-350); the Triton CPU interpreter implements atomics for
            # fp32 only, so low-precision cases carry forward outputs + LSE only.
            try:
                out.backward(tdo)
                rec.update(dq=tq.grad.float().numpy(), dk=tk.grad.float().numpy(),
                           dv=tv.grad.float().numpy())
            except Exception as e:  # noqa: BLE001
                print("   (bwd unavailable in interpreter: %s)" % type(e).__name__)
                bwd = False
        np.savez_compressed(os.path.join(OUT, f"tr_{name}.npz"), **rec)
        print("triton", name, "out", tuple(out.shape), "lse", tuple(lse.shape), "bwd", bwd)


def triton_amd_cases():
    """python/aule/triton_flash_amd.py:97-240 (_flash_attn_fwd_amd, the exp2 kernel the
    reference runs on ROCm) with the autotuner bypassed (it needs a GPU driver)."""
    cases = [
        ("mqa_b1h4kv1sq64sk96d32_causal", 60, 1, 4, 1, 64, 96, 32, True, None, "fp32"),
        ("gqa_b1h8kv2s128d64_causal", 61, 1, 8, 2, 128, 128, 64, True, None, "fp32"),
        ("mha_b1h2s96d128_full_scale", 62, 1, 2, 2, 96, 96, 128, False, 0.25, "fp32"),
        # (a bf16 case is not recorded: the Triton CPU interpreter mis-executes this kernel's
        #  bf16 tl.dot path and returns ~1e8 garbage; bf16 goldens come from the generic kernel)
    ]
    for name, seed, B, Hq, Hkv, Sq, Sk, D, causal, scale, dt in cases:
        q, k, v = make_inputs(seed, B, Hq, Hkv, Sq, Sk, D)
        tdt = TORCH_DT[dt]
        tq, tk, tv = (torch.from_numpy(x).to(tdt).contiguous() for x in (q, k, v))
        out = torch.empty_like(tq)
        L = torch.empty(B, Hq, Sq, dtype=torch.float32)
        sc = (1.0 / math.sqrt(D)) if scale is None else scale
        BM = BN = 64
        grid = (triton.cdiv(Sq, BM), B * Hq)
        ref_amd._flash_attn_fwd_amd.fn[grid](
            tq, tk, tv, out, L,
            *tq.stride(), *tk.stride(), *tv.stride(), *out.stride(), *L.stride(),
            Hq, Hkv, Sq, Sk, D, sc, -1,
            BLOCK_M=BM, BLOCK_N=BN, BLOCK_K=triton.next_power_of_2(D),
            IS_CAUSAL=causal, STORE_LSE=True)
        rec = dict(kind="triton_amd_fwd", seed=seed, shape=np.array([B, Hq, Hkv, Sq, Sk, D]),
                   causal=causal, scale=(-1.0 if scale is None else float(scale)), dtype=dt,
                   q=tq.float().numpy(), k=tk.float().numpy(), v=tv.float().numpy(),
                   out=out.float().numpy(), lse=L.numpy())
        np.savez_compressed(os.path.join(OUT, f"amd_{name}.npz"), **rec)
        print("triton-amd", name, tuple(out.shape))


def triton_amd_window_cases():
    """Sliding window: the same kernel with window_size > 0 -- visible iff
    q_pos - k_pos < window_size, on top of the causal rule (triton_flash_amd.py:179-183)."""
    cases = [  # name, seed, B, Hq, Hkv, Sq, Sk, D, causal, window
        ("causal_w16_b1h2s128d64", 70, 1, 2, 2, 128, 128, 64, True, 16),
        ("causal_w100_gqa_b1h4kv2s200d32", 71, 1, 4, 2, 200, 200, 32, True, 100),
        ("full_w32_mqa_b1h4kv1sq64sk96d128", 72, 1, 4, 1, 64, 96, 128, False, 32),
        ("causal_w1_b1h2s64d64", 73, 1, 2, 2, 64, 64, 64, True, 1),
        ("causal_w70_b2h2s300d128", 74, 2, 2, 2, 300, 300, 128, True, 70),
    ]
    for name, seed, B, Hq, Hkv, Sq, Sk, D, causal, window in cases:
        q, k, v = make_inputs(seed, B, Hq, Hkv, Sq, Sk, D)
        tq, tk, tv = (torch.from_numpy(x).contiguous() for x in (q, k, v))
        out = torch.empty_like(tq)
        L = torch.empty(B, Hq, Sq, dtype=torch.float32)
        sc = 1.0 / math.sqrt(D)
        BM = BN = 64
        grid = (triton.cdiv(Sq, BM), B * Hq)
        ref_amd._flash_attn_fwd_amd.fn[grid](
            tq, tk, tv, out, L,
            *tq.stride(), *tk.stride(), *tv.stride(), *out.stride(), *L.stride(),
            Hq, Hkv, Sq, Sk, D, sc, window,
            BLOCK_M=BM, BLOCK_N=BN, BLOCK_K=triton.next_power_of_2(D),
            IS_CAUSAL=causal, STORE_LSE=True)
        rec = dict(kind="triton_amd_fwd_window", seed=seed, shape=np.array([B, Hq, Hkv, Sq, Sk, D]),
                   causal=causal, scale=-1.0, dtype="fp32", window=window,
                   q=q, k=k, v=v, out=out.numpy(), lse=L.numpy())
        np.savez_compressed(os.path.join(OUT, f"win_{name}.npz"), **rec)
        print("triton-amd window", name, tuple(out.shape))


def paged_cases():
    """Paged-KV decode: flash_attention_paged_amd (triton_flash_amd.py:656-737), interpreted."""
    cases = [  # name, seed, B, Hq, Hkv, D, block_size, num_blocks, context_lens, window, dtype
        ("gqa_b3h8kv2d64_bs16", 80, 3, 8, 2, 64, 16, 24, [37, 100, 1], -1, "fp16"),
        ("mqa_b2h4kv1d128_bs32_w20", 81, 2, 4, 1, 128, 32, 12, [150, 64], 20, "fp16"),
        ("mha_b2h2kv2d32_bs8", 82, 2, 2, 2, 32, 8, 40, [129, 17], -1, "fp16"),
    ]
    for name, seed, B, Hq, Hkv, D, bs, nb, lens, window, dt in cases:
        rng = np.random.RandomState(seed)
        tdt = TORCH_DT[dt]
        q = torch.from_numpy(rng.randn(B, Hq, D).astype(np.float32)).to(tdt)
        kc = torch.from_numpy(rng.randn(nb, bs, Hkv, D).astype(np.float32)).to(tdt)
        vc = torch.from_numpy(rng.randn(nb, bs, Hkv, D).astype(np.float32)).to(tdt)
        max_blocks = (max(lens) + bs - 1) // bs
        bt = np.zeros((B, max_blocks), dtype=np.int32)
        perm = rng.permutation(nb)
        used = 0
        for b in range(B):
            n = (lens[b] + bs - 1) // bs
            bt[b, :n] = perm[used:used + n]
            used += n
        assert used <= nb
        out = ref_amd.flash_attention_paged_amd(q, kc, vc, torch.from_numpy(bt), torch.tensor(lens, dtype=torch.int32),
                                                scale=None, window_size=window)
        rec = dict(kind="triton_amd_paged", seed=seed, dtype=dt, window=window, block_size=bs,
                   q=q.float().numpy(), k_cache=kc.float().numpy(), v_cache=vc.float().numpy(),
                   block_tables=bt, context_lens=np.array(lens, dtype=np.int32), out=out.float().numpy())
        np.savez_compressed(os.path.join(OUT, f"paged_{name}.npz"), **rec)
        print("paged", name, tuple(out.shape))


def rope_cases():
    """RoPE.  The reference's fused kernel (triton_flash.py:112-131, :165-180) does not run --
    it slices register tensors (`q[:, :half_k]`) and builds `tl.arange(0, BLOCK_K // 2)` from a non-constexpr, which
    Triton rejects on any backend -- so the fixtures record what its own self-test (:788-806) holds the fused path
    to: the tables of precompute_rope_frequencies (:644-677), the half-split rotation apply_rope_separate
    (:680-703; pairs (i, i + D/2), position = sequence index) and attention over the rotated Q, K by the
    reference's FA-2 kernel (:386-476, interpreted)."""
    from aule.triton_flash import precompute_rope_frequencies, apply_rope_separate
    cases = [  # name, seed, B, Hq, Hkv, S, D, causal, table_len
        ("mha_b2h8s64d64_causal", 90, 2, 8, 8, 64, 64, True, 64),        # triton_flash.py:808
        ("mha_b1h8s32d128_causal", 91, 1, 8, 8, 32, 128, True, 32),      # :809
        ("mha_b1h4s64d64_full", 92, 1, 4, 4, 64, 64, False, 64),         # :810
        ("gqa_b1h4kv2s100d32_full", 93, 1, 4, 2, 100, 32, False, 128),
        ("mqa_b1h4kv1s70d64_causal", 94, 1, 4, 1, 70, 64, True, 96),
    ]
    for name, seed, B, Hq, Hkv, S, D, causal, tlen in cases:
        q, k, v = make_inputs(seed, B, Hq, Hkv, S, S, D)
        tq, tk, tv = (torch.from_numpy(x) for x in (q, k, v))
        cos, sin = precompute_rope_frequencies(tlen, D, device="cpu")
        qr, kr = apply_rope_separate(tq, tk, cos, sin)
        out = FlashAttentionTritonFunc.apply(qr.contiguous(), kr.contiguous(), tv, causal, None, -1, None, None)
        rec = dict(kind="triton_rope", seed=seed, shape=np.array([B, Hq, Hkv, S, S, D]), causal=causal,
                   scale=-1.0, dtype="fp32", q=q, k=k, v=v, cos=cos.numpy(), sin=sin.numpy(),
                   q_rot=qr.numpy(), k_rot=kr.numpy(), out=out.numpy())
        np.savez_compressed(os.path.join(OUT, f"rope_{name}.npz"), **rec)
        print("rope", name, tuple(out.shape))


if __name__ == "__main__":
    import sys
    if len(sys.argv) > 1 and sys.argv[1] == "rope":
        rope_cases()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "paged":
        paged_cases()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "window":   # add the window fixtures without touching the others
        triton_amd_window_cases()
        sys.exit(0)
    numpy_cases()
    triton_cases()
    triton_amd_cases()
    triton_amd_window_cases()
    paged_cases()
    rope_cases()
