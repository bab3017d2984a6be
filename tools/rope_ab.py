#!/usr/bin/env python3
"""RoPE + attention forward, three ways (fused vs separate, measured):
  two-pass  : rope(Q), rope(K), attention                      (training path; what round 1 shipped)
  q-fused   : rope(K), attention with Q rotated in its registers (aule_attention_forward_rope_ex; inference path)
  k-cached  : attention with Q rotated in registers, K rotated earlier (KV-cache serving: the K pass is paid at append time)
and the plain attention and the two passes alone as the floor / the price list.  Every leg is conditioned on its own
workload (~250 ms of back-to-back launches: MI355X's clock transient), then timed over N launches with
one event pair; TFLOP/s count the attention FLOPs only (the rotations are overhead).
Round 4: the one-wave-per-SIMD forward rotates Q itself (bit-identical to the pass); the round-3 "ps" leg (AULE_HIP_FWD_KERNEL=ps,
the tile-stream kernel) is gone with that kernel -- the switch only knows pp / w4 now and says so on stderr for anything else."""
import math, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "aule-attention_amd"))
import torch
from aule import _torch as at


def timed(fn, n=30, cond_ms=250.0):
    fn(); torch.cuda.synchronize()
    t0 = time.time()
    while (time.time() - t0) * 1e3 < cond_ms:
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3   # us


def run(dtype, B, Hq, Hkv, S, D, causal):
    dt = {"bf16": torch.bfloat16, "fp16": torch.float16}[dtype]
    q = torch.randn(B, Hq, S, D, device="cuda", dtype=dt); k = torch.randn(B, Hkv, S, D, device="cuda", dtype=dt); v = torch.randn_like(k)
    inv = 1.0 / (10000.0 ** (torch.arange(0, D, 2, device="cuda", dtype=torch.float32) / D))
    ang = torch.arange(S, device="cuda", dtype=torch.float32)[:, None] * inv[None, :]
    cos, sin = ang.cos().contiguous(), ang.sin().contiguous()
    sc = 1 / math.sqrt(D)
    code = 1 if causal else 0
    fus = at.rope_fusable(q, k, code, -1, cos, sin, 0)
    kr = at.rope_raw(k, cos, sin); qr = at.rope_raw(q, cos, sin)
    a = at.fwd_raw(qr, kr, v, code, sc, want_lse=False)[0]
    same = None
    if fus:
        b = at.fwd_raw(q, kr, v, code, sc, want_lse=False, q_rope=(cos, sin, 0))[0]
        same = bool(torch.equal(a, b))
    qb, kb = torch.empty_like(q), torch.empty_like(k)
    legs = [
        ("attention only", lambda: at.fwd_raw(qr, kr, v, code, sc, want_lse=False)),
        ("rope(Q) pass alone", lambda: at.rope_raw(q, cos, sin, out=qb)),
        ("rope(K) pass alone", lambda: at.rope_raw(k, cos, sin, out=kb)),
        ("two-pass: rope(Q)+rope(K)+attention", lambda: at.fwd_raw(at.rope_raw(q, cos, sin, out=qb), at.rope_raw(k, cos, sin, out=kb), v, code, sc, want_lse=False)),
        ("k-cached two-pass: rope(Q)+attention", lambda: at.fwd_raw(at.rope_raw(q, cos, sin, out=qb), kr, v, code, sc, want_lse=False)),
    ]
    if fus:
        legs += [
            ("q-fused: rope(K)+attention[Q rotated in registers]", lambda: at.fwd_raw(q, at.rope_raw(k, cos, sin, out=kb), v, code, sc, want_lse=False, q_rope=(cos, sin, 0))),
            ("k-cached: attention[Q rotated in registers]", lambda: at.fwd_raw(q, kr, v, code, sc, want_lse=False, q_rope=(cos, sin, 0))),
        ]
    fl = 4.0 * B * Hq * D * (S * (S + 1) / 2 if causal else S * S)
    print(f"{dtype} B{B} Hq{Hq} Hkv{Hkv} S{S} D{D} causal={int(causal)}   fusable here: {fus}   fused == two-pass bit for bit: {same}", flush=True)
    res = {}
    for name, fn in legs:
        us = timed(fn)
        res[name] = us
        print(f"    {name:54s} {us:9.1f} us   {fl / us / 1e6:7.1f} TF", flush=True)
    if fus:
        tp, qf = res[legs[3][0]], res[legs[5][0]]
        print(f"    q-fused vs two-pass: {(tp / qf - 1) * 100:+.1f} %   (bytes saved: one read + one write of Q = {2 * q.numel() * 2 / 1e6:.0f} MB)", flush=True)


print("lib:", os.environ.get("AULE_LIBRARY_PATH", "(in-tree)"), " AULE_HIP_FWD_KERNEL =", os.environ.get("AULE_HIP_FWD_KERNEL", "(default)"))
run("bf16", 4, 32, 32, 2048, 128, True)     # C2-like, MHA
run("bf16", 4, 32, 8, 2048, 128, True)      # C3-like, GQA 4:1
run("bf16", 2, 32, 8, 8192, 128, True)      # long sequence
run("bf16", 4, 32, 32, 2048, 128, False)
run("fp16", 8, 16, 16, 2048, 64, True)      # D = 64
