#!/usr/bin/env python3
"""Randomised differential test of the whole forward/backward dispatch against the fp64 oracle: random dtype, batch,
GQA ratio, Sq, Sk, head_dim, causal mode, window and scale sign -- the combinations nobody wrote a case for.
Deterministic per seed; prints every failing configuration.   python tools/fuzz_parity.py [n=200] [seed=0]
Other kinds: `paged`, `rope`, `split` (small grids + fused rotation), `long` (round 6: Sq 2048 / 4096, gradients judged head by head),
`window` (round 6: the window instances of the one-wave-per-SIMD forward -- forward + LSE of every row)."""
import ctypes, math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "aule-attention_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import oracle
from aule import _torch as at, _capi
from util import BWD_TOL, LSE_TOL, fwd_tol, quantize, torch_dtype

def route(dtype, B, Hq, Hkv, Sq, Sk, D, code, W):
    d = _capi.AttnDesc(); d.struct_size = ctypes.sizeof(_capi.AttnDesc)
    d.dtype = {"fp32": 0, "fp16": 1, "bf16": 2}[dtype]
    d.batch, d.heads_q, d.heads_kv, d.seq_q, d.seq_k, d.head_dim = B, Hq, Hkv, Sq, Sk, D
    d.causal, d.window_size = code, W
    return _capi.get_lib().aule_hip_debug_forward_route(ctypes.byref(d))

def draw(rng):
    dtype = rng.choice(["bf16", "bf16", "fp16", "fp32"])
    D = int(rng.choice([32, 64, 128]))
    Hkv = int(rng.choice([1, 1, 2, 3, 4, 8])); g = int(rng.choice([1, 1, 2, 4, 8, 5])); Hq = Hkv * g
    B = int(rng.choice([1, 1, 2, 3, 9]))
    Sq = int(rng.choice([1, 1, 2, 3, 7, 16, 17, 31, 33, 64, 65, 100, 200, 256, 257, 300, 512, 700, 1024, 1300]))   # (>= 512: several Q blocks -> the tile stream's seams)
    Sk = int(rng.choice([1, 5, 63, 64, 65, 130, 500, 1023, 1024, 1025, 2047, 2100, 3000, 4096, 5000, 9000]))
    causal = rng.choice(["none", "none", "top", "br", "br"])
    if causal == "br" and Sk < Sq: Sk = Sq + int(rng.choice([0, 1, 100, 1500, 4000]))
    W = int(rng.choice([-1, -1, -1, 1, 7, 64, 100, 1000]))
    scale = None if rng.rand() < 0.7 else float(rng.choice([0.3, -0.2, 0.05, 1.0]))
    # keep the fp64 judge (fwd + bwd ~ 8 B Hq Sq Sk D flops) within ~0.5 s
    while 8.0 * B * Hq * Sq * Sk * D > 6e8:
        if B > 1: B = 1
        elif Hq > Hkv and g > 1: g = max(1, g // 2); Hq = Hkv * g
        elif Hkv > 1: Hkv = 1; Hq = g
        elif Sq > 64 and Sq * 4 > Sk: Sq = Sq // 2
        elif Hq > Hkv and g > 1: g = max(1, g // 2); Hq = Hkv * g
        else: Sk = max(Sq if causal == "br" else 1, Sk // 2)
    return dtype, B, Hq, Hkv, Sq, Sk, D, causal, W, scale

def run(cfg, seed):
    dtype, B, Hq, Hkv, Sq, Sk, D, causal, W, scale = cfg
    cz = {"none": False, "top": True, "br": "bottom-right"}[causal]
    code = {"none": 0, "top": 1, "br": 2}[causal]
    rng = np.random.RandomState(seed)
    q, k, v, do = (quantize(rng.randn(*s).astype(np.float32), dtype) for s in ((B, Hq, Sq, D), (B, Hkv, Sk, D), (B, Hkv, Sk, D), (B, Hq, Sq, D)))
    sc = (1 / math.sqrt(D)) if scale is None else scale
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to("cuda", torch_dtype(dtype))
    tq, tk, tv, tdo = dev(q), dev(k), dev(v), dev(do)
    out, lse = at.fwd_raw(tq, tk, tv, cz, sc, window=W)
    dq, dk, dv = at.bwd_raw(tq, tk, tv, out, tdo, lse, cz, sc, window=W)
    torch.cuda.synchronize()
    ref, rl = oracle.fwd_f64(q, k, v, cz, scale, W)
    rq, rk, rv = oracle.bwd_f64(q, k, v, do, cz, scale, W)
    errs = []
    o = out.float().cpu().numpy()
    atol, rtol = fwd_tol(dtype, float(np.abs(v).max()))
    # fp32 arithmetic error grows with the logit magnitude (dO ~ 2^-23 |S| |V| through exp): a plain NumPy fp32
    # softmax(QK^T)V is 2e-7 from the fp64 judge at |S| ~ 5 and 1.5e-5 at |S| ~ 53 on the same inputs, like the kernel.
    # The suite's fixed 1e-5 assumes the default temperature; here the scale is random, so the bound follows |S|.
    g = Hq // Hkv
    smax = max(float(np.abs(q[b, h] @ k[b, h // g].T).max()) for b in range(B) for h in range(Hq)) * abs(sc)
    grow = 1.0
    if dtype == "fp32":
        atol += 2.0 ** -22 * smax * float(np.abs(v).max())
        grow = max(1.0, smax / 8.0)
    elif smax > 16.0:
        # 16-bit gradients at a temperature far above the default (|S| in the tens: the softmax is nearly one-hot, so the 2^-9
        # roundings of P and dS no longer average over many keys): measured worst over 900 draws 1.04e-2 of max|grad| at
        # scale 1.0, D = 128 (|S| ~ 45) against the suite's 1e-2 for the default temperature -- the bound follows |S| gently
        grow = min(1.5, math.sqrt(smax / 16.0))
    if not np.isfinite(o).all(): errs.append("out non-finite")
    elif (np.abs(o - ref) > atol + rtol * np.abs(ref)).any(): errs.append(f"out err {np.abs(o-ref).max():.3e}")
    gl = lse.cpu().numpy(); fin = np.isfinite(rl)
    if not np.all(np.isneginf(gl[~fin])): errs.append("lse of empty rows not -inf")
    if fin.any() and np.abs(gl[fin] - rl[fin]).max() > LSE_TOL[dtype] + 1e-5 * np.abs(rl[fin]).max(): errs.append(f"lse err {np.abs(gl[fin]-rl[fin]).max():.3e}")
    a, r = BWD_TOL[dtype]
    keys_eff = min(Sk, Sq) if causal == "top" else Sk      # the most keys any row sees
    if W > 0: keys_eff = min(keys_eff, W)
    if dtype == "bf16" and ((scale is not None and abs(scale) * math.sqrt(D) > 2.0) or keys_eff < 32):
        a = 1e-2      # sharp softmax -- a large scale, or so few keys that single weights are O(1) -- (tests/test_gpu_bwd.py, grad_close): the reference's own bar
    for name, got, want in (("dq", dq, rq), ("dk", dk, rk), ("dv", dv, rv)):
        gg = got.float().cpu().numpy()
        if not np.isfinite(gg).all(): errs.append(name + " non-finite"); continue
        # dK / dV accumulate g * Sq rows; where the true gradient vanishes (one key: P = 1, dS = dP - delta cancels
        # exactly) what is left is rounding noise ~ eps |dP| |q| sqrt(rows): plain NumPy fp32 leaves 1.3e-5 at 1600 rows
        acc = max(1.0, math.sqrt(g * Sq / 256.0)) if (dtype == "fp32" and name != "dq") else 1.0
        if dtype == "bf16":
            # the tightened bf16 bound (5e-3 of max|grad|, round 5) was measured on sums over <= ~1000 terms; the 16-bit roundings of dS / P add up like a
            # random walk over the terms a gradient element sums (dQ: Sk keys; dK, dV: g Sq rows): B1 H1 Sq200 Sk5000 D32 reads 6.8e-3 of max|grad| for dQ
            acc = max(1.0, math.sqrt((Sk if name == "dq" else g * Sq) / 512.0))
        tol = a * grow * acc * max(1.0, float(np.abs(want).max()))
        if (np.abs(gg - want) > tol + r * np.abs(want)).any(): errs.append(f"{name} err {np.abs(gg-want).max():.3e} (tol {tol:.1e})")
    return route(dtype, B, Hq, Hkv, Sq, Sk, D, code, W), errs

def run_paged(rng, i):
    """Random paged-decode problem against oracle.paged_decode_f64: shuffled block tables, ragged lengths (0 and 1
    included), power-of-two and other block sizes, GQA ratios, window."""
    import aule
    dtype = rng.choice(["bf16", "fp16"])
    D = int(rng.choice([32, 64, 128])); Hkv = int(rng.choice([1, 2, 4, 8])); g = int(rng.choice([1, 2, 4, 8, 16])); Hq = Hkv * g
    B = int(rng.choice([1, 2, 3, 5, 8])); bs = int(rng.choice([1, 3, 8, 16, 24, 32, 33, 64, 100, 128]))
    lens = [int(rng.choice([0, 1, 2, bs, bs + 1, 31, 32, 33, 100, 777, 1500, 4000])) for _ in range(B)]
    W = int(rng.choice([-1, -1, 1, 17, 256]))
    scale = None if rng.rand() < 0.7 else float(rng.choice([0.3, -0.2]))
    nblk = [(n + bs - 1) // bs for n in lens]
    num_blocks = sum(nblk) + int(rng.choice([0, 1, 5]))
    num_blocks = max(num_blocks, 1)
    r2 = np.random.RandomState(5000 + i)
    q = quantize(r2.randn(B, Hq, D).astype(np.float32), dtype)
    kc = quantize(r2.randn(num_blocks, bs, Hkv, D).astype(np.float32), dtype)
    vc = quantize(r2.randn(num_blocks, bs, Hkv, D).astype(np.float32), dtype)
    bt = np.zeros((B, max(max(nblk), 1) + int(rng.choice([0, 2]))), dtype=np.int32)
    perm = r2.permutation(num_blocks); used = 0
    for b in range(B):
        bt[b, :nblk[b]] = perm[used:used + nblk[b]]; used += nblk[b]
    cl = np.array(lens, dtype=np.int32)
    dt = torch_dtype(dtype)
    out = aule.flash_attention_paged_amd(torch.from_numpy(q).to("cuda", dt), torch.from_numpy(kc).to("cuda", dt),
                                         torch.from_numpy(vc).to("cuda", dt), torch.from_numpy(bt).cuda(), torch.from_numpy(cl).cuda(),
                                         scale=scale, window_size=W).float().cpu().numpy()
    ref = oracle.paged_decode_f64(q, kc, vc, bt, cl, scale, W)
    atol, rtol = fwd_tol(dtype, float(np.abs(vc).max()))
    cfg = (dtype, B, Hq, Hkv, D, bs, lens, W, scale)
    if not np.isfinite(out).all(): return cfg, ["non-finite"]
    bad = np.abs(out - ref) > atol + rtol * np.abs(ref)
    return cfg, ([f"err {np.abs(out-ref).max():.3e}"] if bad.any() else [])


def run_rope(rng, i):
    """Random RoPE pass against oracle.rope_f64: dtype, layout, inverse, offset, vector and scalar head dims, in place."""
    dtype = rng.choice(["bf16", "fp16", "fp32"])
    B, H, S = int(rng.choice([1, 2, 3])), int(rng.choice([1, 2, 5, 8])), int(rng.choice([1, 2, 31, 64, 100, 333]))
    D = int(rng.choice([2, 6, 16, 20, 32, 48, 64, 80, 96, 128]))
    layout = rng.choice(["half", "interleaved"]); inverse = bool(rng.rand() < 0.4); off = int(rng.choice([0, 0, 1, 7, 50]))
    r2 = np.random.RandomState(7000 + i)
    x = quantize(r2.randn(B, H, S, D).astype(np.float32), dtype)
    cos, sin = oracle.rope_tables(S + off + int(rng.choice([0, 3])), D)
    want = oracle.rope_f64(x, cos, sin, layout, inverse, off)
    dev = lambda a, d="fp32": torch.from_numpy(np.ascontiguousarray(a)).to("cuda", torch_dtype(d))
    tx = dev(x, dtype)
    got = at.rope_raw(tx, dev(cos), dev(sin), layout, inverse, off)
    at.rope_raw(tx, dev(cos), dev(sin), layout, inverse, off, out=tx)
    tol = {"fp32": 2e-6, "fp16": 2e-3, "bf16": 1.6e-2}[dtype] * max(1.0, float(np.abs(want).max()))
    cfg = (dtype, B, H, S, D, layout, inverse, off)
    errs = []
    if np.abs(got.float().cpu().numpy() - want).max() > tol: errs.append(f"err {np.abs(got.float().cpu().numpy()-want).max():.3e}")
    if not torch.equal(tx, got): errs.append("in place != out of place")
    return cfg, errs


def run_split(rng, i):
    """Small grids with long keys (route 7: pairs of causal Q blocks, or non-causal blocks, cut into pieces + merge) and, on
    the same draw, the fused query rotation against the two-pass form.  Forward only (out + LSE vs the fp64 judge): the
    shapes that reach this route cost the judge seconds each."""
    dtype = rng.choice(["bf16", "fp16"])
    D = int(rng.choice([64, 128])); Hkv = int(rng.choice([1, 2, 4])); g = int(rng.choice([1, 2, 4])); Hq = Hkv * g
    Sq = int(rng.randint(1100, 3200)); causal = rng.choice(["none", "top", "br"])
    Sk = Sq if causal == "top" and rng.rand() < 0.7 else int(Sq + rng.choice([0, 1, 63, 500, 2000]))
    if causal == "none" and rng.rand() < 0.5: Sk = int(rng.randint(2048, 5000))
    scale = None if rng.rand() < 0.7 else float(rng.choice([0.3, -0.2, 0.05]))
    cz = {"none": False, "top": True, "br": "bottom-right"}[causal]
    code = {"none": 0, "top": 1, "br": 2}[causal]
    r = route(dtype, 1, Hq, Hkv, Sq, Sk, D, code, -1)
    cfg = (dtype, 1, Hq, Hkv, Sq, Sk, D, causal, scale)
    r2 = np.random.RandomState(9000 + i)
    q, k, v = (quantize(r2.randn(*s).astype(np.float32), dtype) for s in ((1, Hq, Sq, D), (1, Hkv, Sk, D), (1, Hkv, Sk, D)))
    sc = (1 / math.sqrt(D)) if scale is None else scale
    dev = lambda a, d=dtype: torch.from_numpy(np.ascontiguousarray(a)).to("cuda", torch_dtype(d))
    tq, tk, tv = dev(q), dev(k), dev(v)
    out, lse = at.fwd_raw(tq, tk, tv, cz, sc)
    ref, rl = oracle.fwd_f64(q, k, v, cz, scale, -1)
    errs = []
    atol, rtol = fwd_tol(dtype, float(np.abs(v).max()))
    o = out.float().cpu().numpy()
    if not np.isfinite(o).all(): errs.append("out non-finite")
    elif (np.abs(o - ref) > atol + rtol * np.abs(ref)).any(): errs.append(f"out err {np.abs(o-ref).max():.3e}")
    if np.abs(lse.cpu().numpy() - rl).max() > LSE_TOL[dtype] + 1e-5 * np.abs(rl).max(): errs.append(f"lse err {np.abs(lse.cpu().numpy()-rl).max():.3e}")
    # fused query rotation == rope(Q) pass + attention, bit for bit, whenever both launches take the plain tile stream
    qoff = Sk - Sq if causal == "br" else 0
    cos, sin = oracle.rope_tables(max(Sq + qoff, Sk) + 2, D)
    tc, ts = dev(cos, "fp32"), dev(sin, "fp32")
    if at.rope_fusable(tq, tk, code, -1, tc, ts, qoff, sc):
        fused = at.fwd_raw(tq, tk, tv, cz, sc, want_lse=False, q_rope=(tc, ts, qoff))[0]
        qr = at.rope_raw(tq, tc, ts, "half", False, qoff)
        two = at.fwd_raw(qr, tk, tv, cz, sc, want_lse=False)[0]
        if r == 6:
            if not torch.equal(fused, two): errs.append("fused rotation != two-pass (same kernel)")
        elif (fused.float() - two.float()).abs().max().item() > 2 * atol + 0.02: errs.append("fused rotation far from two-pass")
    return (r, cfg), errs


def run_long(rng, i):
    """Round 6: the sizes the backward's longest streams run -- Sq in {2048, 4096}, 16-bit, D 64 / 128, MHA and GQA,
    every causal mode, causal windows -- forward (sampled rows) AND backward against the fp64 judges.  The whole-problem C judge
    would take minutes per draw here; gradients are checked head by head (oracle.bwd_head_f64: one KV head + its query group at
    full S) on two random (batch, kv-head) units, the forward on 48 sampled rows (oracle.fwd_rows_f64)."""
    dtype = rng.choice(["bf16", "bf16", "fp16"])
    D = int(rng.choice([64, 128, 128])); Hkv = int(rng.choice([1, 2, 4, 8])); g = int(rng.choice([1, 1, 2, 4])); Hq = Hkv * g
    B = int(rng.choice([1, 2, 4]))
    Sq = int(rng.choice([2048, 4096, 4096, 2048 + 77, 4096 - 131]))
    causal = rng.choice(["none", "top", "top", "br"])
    Sk = Sq if rng.rand() < 0.6 else int(rng.choice([2048, 4096, 5000, 8192]))
    if causal == "br" and Sk < Sq: Sk = Sq + int(rng.choice([0, 100, 4000]))
    W = int(rng.choice([-1, -1, -1, 256, 1024])) if causal != "none" else -1
    scale = None if rng.rand() < 0.8 else float(rng.choice([0.05, 0.12]))
    cz = {"none": False, "top": True, "br": "bottom-right"}[causal]
    code = {"none": 0, "top": 1, "br": 2}[causal]
    cfg = (dtype, B, Hq, Hkv, Sq, Sk, D, causal, W, scale)
    sc = (1 / math.sqrt(D)) if scale is None else scale
    gen = torch.Generator(device="cuda").manual_seed(11000 + i)
    dt = torch_dtype(dtype)
    tq, tdo = (torch.randn(B, Hq, Sq, D, device="cuda", dtype=dt, generator=gen) for _ in range(2))
    tk, tv = (torch.randn(B, Hkv, Sk, D, device="cuda", dtype=dt, generator=gen) for _ in range(2))
    out, lse = at.fwd_raw(tq, tk, tv, cz, sc, window=W)
    dq, dk, dv = at.bwd_raw(tq, tk, tv, out, tdo, lse, cz, sc, window=W)
    torch.cuda.synchronize()
    errs = []
    r2 = np.random.RandomState(12000 + i)
    a, r = BWD_TOL[dtype]
    keys_eff = min(Sk, Sq) if causal == "top" else Sk
    if W > 0: keys_eff = min(keys_eff, W)
    for _ in range(2):
        b, hk = int(r2.randint(B)), int(r2.randint(Hkv))
        f = lambda t, hh=slice(hk * g, (hk + 1) * g): t[b, hh].float().cpu().numpy()
        rq, rk, rv = oracle.bwd_head_f64(f(tq), f(tk, hk), f(tv, hk), f(tdo), None, cz, scale, W)
        for name, got, want in (("dq", f(dq), rq), ("dk", f(dk, hk), rk), ("dv", f(dv, hk), rv)):
            if not np.isfinite(got).all(): errs.append(f"{name}[{b},{hk}] non-finite"); continue
            # (the accumulation-length term of run(): a gradient element sums Sk keys (dQ) or g Sq rows (dK, dV) of 16-bit-rounded terms)
            acc = max(1.0, math.sqrt(min(keys_eff if name == "dq" else g * Sq, 8192) / 512.0)) if dtype == "bf16" else 1.0
            tol = a * acc * max(1.0, float(np.abs(want).max()))
            e = np.abs(got - want)
            if (e > tol + r * np.abs(want)).any(): errs.append(f"{name}[{b},{hk}] err {e.max():.3e} = {e.max()/max(1.0, float(np.abs(want).max())):.2e} of max|grad| (tol {tol:.1e})")
    rows = r2.randint(0, B * Hq * Sq, size=48).astype(np.int64)
    qf, kf, vf = (t.float().cpu().numpy() for t in (tq, tk, tv))
    ro, rl = oracle.fwd_rows_f64(qf, kf, vf, rows, cz, scale, W)
    o = out.float().cpu().numpy().reshape(-1, D)[rows]; gl = lse.cpu().numpy().reshape(-1)[rows]
    atol, rtol = fwd_tol(dtype, float(np.abs(vf).max()))
    fin = np.isfinite(rl)
    if (np.abs(o - ro) > atol + rtol * np.abs(ro)).any(): errs.append(f"out err {np.abs(o-ro).max():.3e}")
    if fin.any() and np.abs(gl[fin] - rl[fin]).max() > LSE_TOL[dtype] + 1e-5 * np.abs(rl[fin]).max(): errs.append(f"lse err {np.abs(gl[fin]-rl[fin]).max():.3e}")
    return (route(dtype, B, Hq, Hkv, Sq, Sk, D, code, W), cfg), errs


def run_window(rng, i):
    """Round 6: causal sliding windows on the one-wave-per-SIMD forward's window instances (fa_fwd_w4_gfx950.hip WIN): 16-bit, D 64 / 128, top-left
    and bottom-right causal with offsets that are no tile multiples, windows from 128 keys to beyond the sequence -- aligned and not --, ragged
    last blocks, GQA; one draw in four with a key whose logit towers over its head's, placed at random (in front of some rows' windows, inside
    others': the fixed reference under- or overflows and the exact-maximum stream has to repair those parts).  Every output row and LSE
    against the fp64 judge."""
    dtype = rng.choice(["bf16", "bf16", "fp16"])
    D = int(rng.choice([64, 128, 128])); Hkv = int(rng.choice([1, 2, 4])); g = int(rng.choice([1, 1, 2, 4])); Hq = Hkv * g
    B = int(rng.choice([1, 1, 2]))
    Sq = int(rng.choice([257, 300, 512, 513, 700, 1000, 1024, 1100, 1536, 2048, 2048 + 77, 3000]))
    causal = rng.choice(["top", "top", "br"])
    Sk = Sq if causal == "top" else Sq + int(rng.choice([0, 1, 63, 64, 100, 1000, 2049]))
    W = int(rng.choice([128, 129, 191, 192, 200, 255, 256, 257, 300, 511, 512, 640, 1000, 1024, 1500, 2500]))
    scale = None if rng.rand() < 0.8 else float(rng.choice([0.05, 0.2]))
    while 4.0 * B * Hq * Sq * Sk * D > 1.2e9:       # (the fp64 judge within ~1 s)
        if B > 1: B = 1
        elif g > 1: g //= 2; Hq = Hkv * g
        elif Hkv > 1: Hkv //= 2; Hq = Hkv * g
        else: break
    cz = {"top": True, "br": "bottom-right"}[causal]
    code = {"top": 1, "br": 2}[causal]
    spike = rng.rand() < 0.25
    cfg = (dtype, B, Hq, Hkv, Sq, Sk, D, causal, W, scale, "spike" if spike else "")
    r2 = np.random.RandomState(21000 + i)
    q, k, v = (quantize(r2.randn(*s).astype(np.float32), dtype) for s in ((B, Hq, Sq, D), (B, Hkv, Sk, D), (B, Hkv, Sk, D)))
    if spike:
        row, key = int(r2.randint(Sq)), int(r2.randint(Sk))
        k[:, 0, key, :] = quantize((20.0 * q[:, 0, row, :]).astype(np.float32), dtype)
    sc = (1 / math.sqrt(D)) if scale is None else scale
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to("cuda", torch_dtype(dtype))
    out, lse = at.fwd_raw(dev(q), dev(k), dev(v), cz, sc, window=W)
    torch.cuda.synchronize()
    ref, rl = oracle.fwd_f64(q, k, v, cz, scale, W)
    errs = []
    o = out.float().cpu().numpy(); gl = lse.cpu().numpy()
    atol, rtol = fwd_tol(dtype, float(np.abs(v).max()))
    if not np.isfinite(o).all(): errs.append("out non-finite")
    elif (np.abs(o - ref) > atol + rtol * np.abs(ref)).any(): errs.append(f"out err {np.abs(o-ref).max():.3e}")
    smax = abs(sc) * float(np.abs((q[:, 0, row, :].astype(np.float64) * k[:, 0, key, :]).sum(-1)).max()) if spike else 1.0   # (LSE at |S| in the hundreds: fp32 arithmetic)
    if np.abs(gl - rl).max() > LSE_TOL[dtype] * max(1.0, smax / 64.0) + 1e-5 * np.abs(rl).max(): errs.append(f"lse err {np.abs(gl-rl).max():.3e}")
    return (route(dtype, B, Hq, Hkv, Sq, Sk, D, code, W), cfg), errs


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "one":
        # one configuration again: python tools/fuzz_parity.py one bf16 1 4 2 31 500 128 br -1 0.3 1001   (scale: a number or "none")
        a = sys.argv[2:]
        cfg = (a[0], int(a[1]), int(a[2]), int(a[3]), int(a[4]), int(a[5]), int(a[6]), a[7], int(a[8]), None if a[9] == "none" else float(a[9]))
        r, errs = run(cfg, int(a[10]))
        print(f"route={r} cfg={cfg}: {'; '.join(errs) if errs else 'clean'}")
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] in ("split", "long", "window"):
        kind = sys.argv[1]
        n = int(sys.argv[2]) if len(sys.argv) > 2 else 40; seed = int(sys.argv[3]) if len(sys.argv) > 3 else 0
        rng = np.random.RandomState(seed); bad = 0; routes = {}
        for i in range(n):
            try:
                (r, cfg), errs = {"split": run_split, "long": run_long, "window": run_window}[kind](rng, i)
            except Exception as e:  # noqa: BLE001
                (r, cfg), errs = (-9, ("?",)), [f"EXCEPTION {type(e).__name__}: {str(e)[:160]}"]
            routes[r] = routes.get(r, 0) + 1
            if errs:
                bad += 1; print(f"FAIL {kind} #{i} route={r} cfg={cfg}: {'; '.join(errs)}", flush=True)
        print(f"{kind}: {n} configurations, {bad} failing; forward routes exercised: {dict(sorted(routes.items()))}")
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] in ("paged", "rope"):
        mode = sys.argv[1]; n = int(sys.argv[2]) if len(sys.argv) > 2 else 200; seed = int(sys.argv[3]) if len(sys.argv) > 3 else 0
        rng = np.random.RandomState(seed); bad = 0
        for i in range(n):
            try:
                cfg, errs = (run_paged if mode == "paged" else run_rope)(rng, i)
            except Exception as e:  # noqa: BLE001
                cfg, errs = ("?",), [f"EXCEPTION {type(e).__name__}: {str(e)[:160]}"]
            if errs:
                bad += 1; print(f"FAIL {mode} #{i} cfg={cfg}: {'; '.join(errs)}", flush=True)
        print(f"{mode}: {n} configurations, {bad} failing")
        sys.exit(0)
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    rng = np.random.RandomState(seed)
    routes, bad = {}, 0
    for i in range(n):
        cfg = draw(rng)
        try:
            r, errs = run(cfg, 1000 + i)
        except Exception as e:  # noqa: BLE001
            r, errs = -9, [f"EXCEPTION {type(e).__name__}: {str(e)[:160]}"]
        routes[r] = routes.get(r, 0) + 1
        if errs:
            bad += 1
            print(f"FAIL #{i} route={r} cfg={cfg}: {'; '.join(errs)}", flush=True)
    print(f"{n} configurations, {bad} failing; forward routes exercised: {dict(sorted(routes.items()))}")
