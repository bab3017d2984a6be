#!/usr/bin/env python3
"""Triage and A/B timing of the tiled 16-bit forward kernels: the judge (fp64 reference on the GPU) and the timing loop that
tools/w4_check.py, tools/w4_d64_check.py and the A/B scripts share.  (Round 2 wrote it for the two-waves-per-SIMD stream kernel,
retired in round 4; its own `check` list pins routes that kernel had and is kept for the shapes, not the pins.)

    python tools/fwd_check.py check        parity on shapes that exercise every seam kind, vs an fp64 reference on the GPU
    python tools/fwd_check.py bench [tag]  per-launch times (HIP events) of the headline shapes for the kernel the
                                          environment selects (AULE_HIP_FWD_KERNEL=pp: the predecessor); long warm-up
    python tools/fwd_check.py dvfs         per-launch times of the first 300 launches of C2 on an idle chip

The fp64 reference is plain torch on the GPU (softmax(QK^T)V per sampled head): a tool-side judge for shapes the C oracle
would take minutes on; the tests in tests/ use the oracle.
"""
import ctypes, math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "aule-attention_amd"))
import torch
from aule import _torch as at, _capi

DT = {"bf16": torch.bfloat16, "fp16": torch.float16}
# worst-case unit roundoff of the P and O roundings (tests/util.py budgets HALF of it for P: the typical case.  A one-hot row
# whose key is not at the softmax reference hits the worst case -- DESIGN.md 4, tools/raw_diag.py)
UNIT = {"bf16": 2.0 ** -8, "fp16": 2.0 ** -11}


def route(dtype, B, Hq, Hkv, Sq, Sk, D, causal, scale=1.0):
    lib = _capi.get_lib()
    lib.aule_hip_debug_forward_route.restype = ctypes.c_int32
    lib.aule_hip_debug_forward_route.argtypes = [ctypes.POINTER(_capi.AttnDesc)]
    d = _capi.AttnDesc()
    d.struct_size = ctypes.sizeof(_capi.AttnDesc)
    d.dtype = {"fp16": 1, "bf16": 2}[dtype]
    d.batch, d.heads_q, d.heads_kv, d.seq_q, d.seq_k, d.head_dim = B, Hq, Hkv, Sq, Sk, D
    d.scale = scale
    d.causal = at.causal_code(causal)
    d.window_size = -1
    return lib.aule_hip_debug_forward_route(ctypes.byref(d))


def ref_head(q, k, v, causal, scale):
    """q [Sq,D], k/v [Sk,D] (16-bit device tensors) -> (O, LSE) in fp64; causal: 0 none, 1 top-left, 2 bottom-right."""
    Sq, Sk = q.shape[0], k.shape[0]
    s = (q.double() @ k.double().T) * scale
    if causal:
        coff = Sk - Sq if causal == 2 else 0
        i = torch.arange(Sq, device=q.device)[:, None] + coff
        j = torch.arange(Sk, device=q.device)[None, :]
        s = s.masked_fill(j > i, float("-inf"))
    lse = torch.logsumexp(s, dim=1)
    p = torch.exp(s - lse[:, None])
    return p @ v.double(), lse


def check(dtype, B, Hq, Hkv, Sq, Sk, D, causal, mag=1.0, scale=None, want_route=6, nsample=6, seed=3, qzero=False):
    g = torch.Generator(device="cuda").manual_seed(seed)
    mk = lambda *s: (torch.randn(*s, device="cuda", generator=g) * mag).to(DT[dtype])
    q, k, v = mk(B, Hq, Sq, D), mk(B, Hkv, Sk, D), mk(B, Hkv, Sk, D)
    if qzero:
        q.zero_()   # every logit equal: uniform attention, row sums = number of visible keys
    sc = 1 / math.sqrt(D) if scale is None else scale
    cz = at.causal_code(causal)
    r = route(dtype, B, Hq, Hkv, Sq, Sk, D, causal, sc)
    out, lse = at.fwd_raw(q, k, v, causal, sc)
    out2, _ = at.fwd_raw(q, k, v, causal, sc, want_lse=False)
    torch.cuda.synchronize()
    heads = [(0, 0), (B - 1, Hq - 1)]
    gen = torch.Generator().manual_seed(seed)
    while len(heads) < min(nsample, B * Hq):
        b, h = int(torch.randint(B, (1,), generator=gen)), int(torch.randint(Hq, (1,), generator=gen))
        if (b, h) not in heads:
            heads.append((b, h))
    worst_o = worst_l = 0.0
    bad = 0
    vmax = float(v.float().abs().max())
    for b, h in heads:
        hk = h // (Hq // Hkv)
        ro, rl = ref_head(q[b, h], k[b, hk], v[b, hk], cz, sc)
        eo = (out[b, h].double() - ro).abs()
        bound = 1e-3 + UNIT[dtype] * vmax + UNIT[dtype] * ro.abs()
        bad += int((eo > bound).sum()) + int((~torch.isfinite(out[b, h].float())).sum())
        worst_o = max(worst_o, float(eo.max()))
        worst_l = max(worst_l, float((lse[b, h].double() - rl).abs().max()))
    same = bool(torch.equal(out, out2))
    ok = bad == 0 and worst_l < 1e-3 * max(1.0, mag * mag) and same and (want_route is None or r == want_route)
    print(f"{'ok  ' if ok else 'FAIL'} {dtype} B{B} Hq{Hq} Hkv{Hkv} Sq{Sq} Sk{Sk} D{D} causal={causal} mag={mag} scale={scale}: "
          f"route {r} O err {worst_o:.3e} ({bad} out of tol) LSE err {worst_l:.3e} rerun-identical={same}", flush=True)
    return ok


def run_checks():
    ok = True
    C = [
        # dtype, B, Hq, Hkv, Sq, Sk, D, causal
        ("bf16", 1, 2, 2, 256, 256, 128, True),        # one block, 4 tiles
        ("bf16", 1, 2, 2, 300, 300, 128, True),        # pair, ragged
        ("bf16", 1, 4, 2, 512, 512, 128, True),
        ("bf16", 2, 4, 1, 1024, 1024, 128, True),
        ("bf16", 1, 3, 3, 1280, 1280, 128, True),      # 5 blocks: the middle one unpaired
        ("bf16", 4, 32, 32, 2048, 2048, 128, True),    # 512 items: two per workgroup
        ("bf16", 4, 32, 8, 2048, 2048, 128, True),     # C3 forward (GQA)
        ("bf16", 4, 32, 32, 4096, 4096, 128, True),    # C2: four items = eight parts per workgroup
        ("bf16", 1, 2, 2, 200, 333, 128, False),
        ("bf16", 1, 2, 2, 777, 260, 128, False),
        ("bf16", 1, 64, 64, 1024, 1024, 128, False),
        ("bf16", 2, 40, 8, 1280, 1280, 128, False),    # 400 items
        ("bf16", 4, 32, 32, 4096, 4096, 128, False),
        ("bf16", 1, 8, 8, 1024, 512, 128, True),       # Sq > Sk, top-left
        ("bf16", 1, 8, 8, 512, 1024, 128, "bottom-right"),
        ("bf16", 4, 32, 8, 1000, 3000, 128, "bottom-right"),   # (256 pairs of blocks: one per CU -- stays on route 6)
        ("fp16", 2, 8, 8, 1024, 1024, 128, True),
        ("fp16", 1, 32, 1, 4096, 4096, 64, False),     # C5-like (MQA, D64)
        ("fp16", 8, 32, 32, 2048, 2048, 64, True),
        ("bf16", 8, 32, 32, 2048, 2048, 64, True),     # 1024 items
        ("bf16", 2, 8, 8, 1024, 1024, 32, True),
        ("fp16", 2, 8, 8, 640, 900, 32, False),
        ("bf16", 64, 64, 64, 2304, 2304, 32, True),    # 20480 items: > 64 per CU -> two workgroups per CU
    ]
    for c in C:
        ok &= check(*c)
    ok &= check("bf16", 1, 2, 2, 512, 512, 128, True, mag=6.0)      # fixed-reference range fails -> second stream
    ok &= check("bf16", 4, 32, 32, 2048, 2048, 128, True, mag=5.0)  # ... in the middle of long part lists
    ok &= check("bf16", 1, 2, 2, 512, 512, 128, False, mag=12.0)
    ok &= check("fp16", 1, 2, 2, 512, 512, 128, True, mag=6.0)      # fp16 fixed reference: weights overflow -> second stream
    ok &= check("fp16", 2, 8, 2, 1024, 1024, 64, False, mag=4.0)
    ok &= check("fp16", 16, 16, 16, 512, 33024, 64, False, qzero=True)  # row sums 33024 > 2^15 without any overflow: verdict fails, still exact
    ok &= check("bf16", 2, 4, 4, 1024, 1024, 128, True, scale=-0.1)
    ok &= check("bf16", 2, 4, 4, 1024, 1024, 64, False, scale=0.3)
    ok &= check("bf16", 2, 8, 2, 1000, 3000, 128, "bottom-right", want_route=7)   # few pairs, long keys: the split instances
    ok &= check("bf16", 1, 8, 8, 4096, 4096, 128, False, want_route=7)
    # shapes that must stay on the predecessor (fewer than 4 tiles in the first part)
    ok &= check("bf16", 1, 2, 2, 64, 64, 128, True, want_route=1)
    ok &= check("bf16", 1, 2, 2, 777, 130, 128, False, want_route=1)
    print("ALL OK" if ok else "SOME FAILED", flush=True)
    return ok


def bench_shape(dtype, B, Hq, Hkv, S, D, causal, lse=True, warm=120, iters=100):
    q = torch.randn(B, Hq, S, D, device="cuda", dtype=DT[dtype])
    k = torch.randn(B, Hkv, S, D, device="cuda", dtype=DT[dtype])
    v = torch.randn(B, Hkv, S, D, device="cuda", dtype=DT[dtype])
    sc = 1 / math.sqrt(D)
    for _ in range(warm):
        at.fwd_raw(q, k, v, causal, sc, want_lse=lse)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(iters + 1)]
    ev[0].record()
    for i in range(iters):
        at.fwd_raw(q, k, v, causal, sc, want_lse=lse)
        ev[i + 1].record()
    torch.cuda.synchronize()
    ts = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(iters))
    mean = ev[0].elapsed_time(ev[iters]) / iters
    fl = 4.0 * B * Hq * D * (S * (S + 1) / 2 if causal else S * S)
    tf = lambda ms: fl / ms / 1e9
    print(f"  {dtype} B{B} Hq{Hq} Hkv{Hkv} S{S} D{D} causal={int(bool(causal))} lse={int(lse)}: mean {mean*1e3:.1f} us = {tf(mean):.0f} TF | "
          f"median {ts[iters//2]*1e3:.1f} us = {tf(ts[iters//2]):.0f} TF | min {ts[0]*1e3:.1f} max {ts[-1]*1e3:.1f}", flush=True)


def run_bench(tag):
    print(f"bench [{tag}] kernel={os.environ.get('AULE_HIP_FWD_KERNEL', '(default)')} route C2={route('bf16', 4, 32, 32, 4096, 4096, 128, True)}", flush=True)
    bench_shape("bf16", 4, 32, 32, 4096, 128, True)
    bench_shape("bf16", 4, 32, 32, 4096, 128, True, lse=False)
    bench_shape("bf16", 4, 32, 32, 4096, 128, False)
    bench_shape("bf16", 4, 32, 8, 2048, 128, True)
    bench_shape("bf16", 8, 32, 32, 8192, 128, True, warm=20, iters=20)
    bench_shape("fp16", 1, 32, 1, 16384, 64, False, warm=30, iters=30)
    bench_shape("bf16", 1, 8, 8, 8192, 128, True)
    bench_shape("bf16", 16, 16, 16, 1024, 128, True)


def run_dvfs():
    q = torch.randn(4, 32, 4096, 128, device="cuda", dtype=torch.bfloat16)
    k = torch.randn_like(q); v = torch.randn_like(q)
    sc = 1 / math.sqrt(128)
    torch.cuda.synchronize()
    import time
    for idle_s in (0.0, 2.0):
        time.sleep(idle_s)
        n = 300
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
        ev[0].record()
        for i in range(n):
            at.fwd_raw(q, k, v, True, sc)
            ev[i + 1].record()
        torch.cuda.synchronize()
        ts = [ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(n)]
        print(f"dvfs after {idle_s:.0f} s idle, us per launch:", " ".join(f"{t:.0f}" for t in ts), flush=True)


if __name__ == "__main__":
    mode = sys.argv[1] if len(sys.argv) > 1 else "check"
    if mode == "check":
        sys.exit(0 if run_checks() else 1)
    elif mode == "mag":   # the large-logit cases alone (run with AULE_HIP_FWD_KERNEL=pp / AULE_HIP_FWD_SOFTMAX=classic to compare)
        check("bf16", 1, 2, 2, 512, 512, 128, True, mag=6.0, want_route=None)
        check("bf16", 4, 32, 32, 2048, 2048, 128, True, mag=5.0, want_route=None)
        check("bf16", 1, 2, 2, 512, 512, 128, False, mag=12.0, want_route=None)
        check("fp16", 1, 2, 2, 512, 512, 128, True, mag=6.0, want_route=None)
    elif mode == "one":   # one shape, short: for rocprofv3 passes (tools/pmc_ab.sh)
        dt, B, Hq, Hkv, S, D, causal, it = sys.argv[2], *[int(x) for x in sys.argv[3:10]]
        bench_shape(dt, B, Hq, Hkv, S, D, bool(causal), warm=5, iters=it)
    elif mode == "bench":
        run_bench(sys.argv[2] if len(sys.argv) > 2 else "")
    elif mode == "dvfs":
        run_dvfs()
