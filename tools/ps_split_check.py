#!/usr/bin/env python3
"""Small causal grids: the tile stream with every pair of Q blocks cut in two (route 7) against the plain stream (route 6,
AULE_HIP_FWD_PSSPLIT=0 -- read once per process, so each arm is a process of its own: run this file twice).
Prints conditioned per-launch times; `check` also compares sampled rows with the fp64 oracle."""
import math, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "aule-attention_amd"))
import torch
from aule import _torch as at

SHAPES = [  # dtype, B, Hq, Hkv, S, D  (+ "nc": non-causal)
    ("bf16", 1, 8, 8, 8192, 128),      # the judge's shape: 128 paired items
    ("bf16", 1, 32, 8, 2048, 128),     # single-sequence prefill, 128 paired items
    ("bf16", 1, 16, 16, 4096, 128),    # 128 paired items
    ("bf16", 1, 8, 8, 4096, 128),      # 64 paired items: split -> 128
    ("bf16", 2, 8, 8, 8192, 128),      # 256 paired items: not split (control)
    ("bf16", 1, 32, 32, 1024, 128),    # 64 paired items of 20 tiles
    ("bf16", 1, 8, 8, 1024, 128),      # 16 items
    ("bf16", 1, 8, 8, 512, 128),       # 8 items of 12 tiles: is the extra launch still paid for?
    ("bf16", 4, 8, 8, 512, 128),
    ("fp16", 1, 32, 32, 2048, 64),     # D = 64: 128 items on 512 slots
    ("fp16", 1, 64, 8, 4096, 64),      # D = 64: 512 paired items (control)
    ("bf16", 1, 8, 8, 4096, 128, "nc"),    # non-causal: 128 single blocks of 64 tiles
    ("bf16", 1, 8, 8, 2048, 128, "nc"),    # 64 blocks of 32 tiles
    ("bf16", 1, 16, 16, 4096, 128, "nc"),  # 256 blocks (control)
]


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
    return e0.elapsed_time(e1) / n * 1e3


if len(sys.argv) > 1:
    SHAPES = [SHAPES[int(x)] for x in sys.argv[1:]]   # only these rows (profiling runs)
print("AULE_HIP_FWD_PSSPLIT =", os.environ.get("AULE_HIP_FWD_PSSPLIT", "(unset: on)"), flush=True)
for dtype, B, Hq, Hkv, S, D, *rest in SHAPES:
    causal = 0 if rest else 1
    dt = {"bf16": torch.bfloat16, "fp16": torch.float16}[dtype]
    q = torch.randn(B, Hq, S, D, device="cuda", dtype=dt); k = torch.randn(B, Hkv, S, D, device="cuda", dtype=dt); v = torch.randn_like(k)
    sc = 1 / math.sqrt(D)
    us = timed(lambda: at.fwd_raw(q, k, v, causal, sc, want_lse=True))
    fl = 4.0 * B * Hq * D * (S * (S + 1) / 2 if causal else S * S)
    print(f"  {dtype} B{B} Hq{Hq} Hkv{Hkv} S{S} D{D} {'causal' if causal else 'non-causal'}: {us:9.1f} us  {fl / us / 1e6:7.1f} TF", flush=True)
