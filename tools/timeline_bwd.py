#!/usr/bin/env python3
"""Phase timeline of the dK/dV backward kernel (bf16 D128 causal): s_memtime stamps of workgroup 0 from the TL build
(aule_hip_debug_backward_timeline), six per 32-row query tile and wave:
   0 loop top (next tile's loads issued)  1 S/dP MFMAs retired  2 P/dS arithmetic done  3 dV/dK MFMAs retired
   4 next tile written to LDS             5 barrier passed
Prints, per wave and over the steady-state tiles, the median length of each phase in shader cycles next to its ideal
(16 MFMAs of 32x32x16 = 512 cycles per MFMA phase when the wave has the SIMD's matrix pipe to itself; two waves share a
SIMD, so 1024 per phase if both are in an MFMA phase at once)."""
import ctypes, math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "aule-attention_amd"))
import numpy as np
# the timeline hooks live in the debug library only (cd aule-attention_amd/csrc && make dbg)
os.environ.setdefault("AULE_LIBRARY_PATH", os.path.join(ROOT, "build", "variants", "libaule_dbg.so"))
import torch
from aule import _capi, _torch as at

B, Hq, Hkv, S, D = 4, 32, 8, 2048, 128
NW, MAX = 8, 384
torch.manual_seed(0)
q = torch.randn(B, Hq, S, D, device="cuda", dtype=torch.bfloat16)
k = torch.randn(B, Hkv, S, D, device="cuda", dtype=torch.bfloat16); v = torch.randn_like(k); do = torch.randn_like(q)
sc = 1 / math.sqrt(D)
out, lse = at.fwd_raw(q, k, v, True, sc)
lib = _capi.get_lib()
d = _capi.AttnBwdDesc(); d.struct_size = ctypes.sizeof(_capi.AttnBwdDesc); d.dtype = 2
d.batch, d.heads_q, d.heads_kv, d.seq_q, d.seq_k, d.head_dim = B, Hq, Hkv, S, S, D
d.scale, d.causal, d.window_size, d.device = sc, 1, -1, 0
d.stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
d.q, d.k, d.v, d.out, d.dout, d.lse = q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), do.data_ptr(), lse.data_ptr()
d.dq, d.dk, d.dv = dq.data_ptr(), dk.data_ptr(), dv.data_ptr()
n = int(lib.aule_attention_backward_workspace_size(ctypes.byref(d)))
ws = torch.empty(n, device="cuda", dtype=torch.uint8); d.workspace, d.workspace_bytes = ws.data_ptr(), n
stamps = torch.zeros(NW * MAX, device="cuda", dtype=torch.int64)
DQ = len(sys.argv) > 1 and sys.argv[1] == "dq"
fn = lib.aule_hip_debug_backward_timeline_dq if DQ else lib.aule_hip_debug_backward_timeline
fn.restype = ctypes.c_int32; fn.argtypes = [ctypes.POINTER(_capi.AttnBwdDesc), ctypes.c_void_p]
for _ in range(3):
    rc = fn(ctypes.byref(d), ctypes.c_void_p(stamps.data_ptr()))
    assert rc == 0, rc
torch.cuda.synchronize()
# the instrumented kernel still computes the right thing
rq, rk, rv = at.bwd_raw(q, k, v, out, do, lse, True, sc)
print("TL build == production kernels: dQ %s dK %s dV %s" % (torch.equal(dq, rq), torch.equal(dk, rk), torch.equal(dv, rv)))
t = stamps.cpu().numpy().reshape(NW, MAX).astype(np.int64)
if DQ:
    # dQ kernel: 8 stamps per 64-key tile; waves 0-3 = group 0, waves 4-7 = group 1 (one phase behind)
    nt8 = MAX // 8
    t = t[:, :nt8 * 8].reshape(NW, nt8, 8)
    nm = ["0>1 write next K/V tile to LDS", "1>2 issue loads", "2>3 dS arithmetic", "3>4 barrier wait (end of V-phase)",
          "4>5 dQ MFMAs (16: ideal 512)", "5>6 S/dP MFMAs of next tile (32: ideal 1024)", "6>7 barrier wait (end of M-phase)", "7>0' (loop)"]
    lo, hi = 6, nt8 - 3
    print(f"dQ kernel, tiles {lo}..{hi-1} of workgroup 0; cycles, median per wave (waves 0-3 group 0, 4-7 group 1)")
    for ph in range(8):
        dl = (t[:, lo:hi, ph + 1] - t[:, lo:hi, ph]) if ph < 7 else (t[:, lo + 1:hi + 1, 0] - t[:, lo:hi, 7])
        med = np.median(dl, axis=1)
        print(f"  {nm[ph]:46s} " + " ".join(f"{int(m):5d}" for m in med) + f"   | all: {int(np.median(dl))} [{int(np.percentile(dl,10))} .. {int(np.percentile(dl,90))}]")
    per = np.median(t[:, lo + 1:hi + 1, 0] - t[:, lo:hi, 0], axis=1)
    print("  tile period                                    " + " ".join(f"{int(m):5d}" for m in per) + f"   | own MFMAs (48 x 32 = 1536) / period = {1536/np.median(per):.2f}; two groups per SIMD: {2*1536/np.median(per):.2f}")
    g0v = t[0, lo:hi, 0]; g1v = t[4, lo:hi, 0]
    print("  group 1's V-phase starts this many cycles after group 0's (median):", int(np.median(g1v - g0v)))
    sys.exit(0)
ntile = MAX // 6
t = t[:, :ntile * 6].reshape(NW, ntile, 6)
names = ["0>1 S/dP MFMA phase (ideal 512..1024)", "1>2 P/dS arithmetic", "2>3 dV/dK MFMA phase (ideal 512..1024)",
         "3>4 write next tile to LDS", "4>5 barrier wait", "5>0' loads for tile+2 issued"]
lo, hi = 12, ntile - 2     # steady state: past the diagonal tiles of key block 0
print(f"tiles {lo}..{hi-1} of workgroup 0 (key block 0, all waves active); cycles, median [p10 .. p90] per wave")
tot = np.zeros(NW)
for ph in range(6):
    if ph < 5: dlt = t[:, lo:hi, ph + 1] - t[:, lo:hi, ph]
    else: dlt = t[:, lo + 1:hi + 1, 0] - t[:, lo:hi, 5]
    med = np.median(dlt, axis=1); tot += med
    print(f"  {names[ph]:40s} " + " ".join(f"{int(m):5d}" for m in med) + f"   | all waves: {int(np.median(dlt))} [{int(np.percentile(dlt,10))} .. {int(np.percentile(dlt,90))}]")
per_tile = np.median(t[:, lo + 1:hi + 1, 0] - t[:, lo:hi, 0], axis=1)
print("  tile period (stamp 0 to next stamp 0)    " + " ".join(f"{int(m):5d}" for m in per_tile) + f"   | MFMA share if alone: {1024/np.median(per_tile):.2f}")
print("  skew between waves at the barrier exit (max-min of stamp 5), median over tiles:", int(np.median(t[:, lo:hi, 5].max(0) - t[:, lo:hi, 5].min(0))))
print("  arrival spread at the barrier (max-min of stamp 4), median:", int(np.median(t[:, lo:hi, 4].max(0) - t[:, lo:hi, 4].min(0))))
