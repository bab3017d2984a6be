#!/usr/bin/env python3
"""Cycle accounting of the one-wave-per-SIMD dK/dV kernel (fa_bwd_dkv4_gfx950.hip, timeline build: debug library, AULE_TL=dkv4):
per wave of workgroup 0 the number of stream iterations and the shader cycles spent in [phase 1 + phase boundary] and in
[phase 2], measured at points where the wave has just waited for its LDS reads anyway.
    python tools/timeline_dkv4.py [causal] [B] [Hq] [Hkv] [S] [D = 128 | 64]"""
import ctypes, math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "aule-attention_amd"))
os.environ.setdefault("AULE_LIBRARY_PATH", os.path.join(ROOT, "build", "variants", "libaule_dbg.so"))
os.environ["AULE_TL"] = "dkv4"
import torch
from aule import _capi, _torch as at
causal = int(sys.argv[1]) if len(sys.argv) > 1 else 0
B, Hq, Hkv, S = (int(x) for x in sys.argv[2:6]) if len(sys.argv) > 5 else (2, 16, 16, 4096)
D = int(sys.argv[6]) if len(sys.argv) > 6 else 128
q = torch.randn(B, Hq, S, D, device="cuda", dtype=torch.bfloat16)
k = torch.randn(B, Hkv, S, D, device="cuda", dtype=torch.bfloat16); v = torch.randn_like(k); do = torch.randn_like(q)
sc = 1 / math.sqrt(D)
out, lse = at.fwd_raw(q, k, v, causal, sc)
lib = _capi.get_lib()
d = _capi.AttnBwdDesc(); d.struct_size = ctypes.sizeof(_capi.AttnBwdDesc); d.dtype = 2
d.batch, d.heads_q, d.heads_kv, d.seq_q, d.seq_k, d.head_dim = B, Hq, Hkv, S, S, D
d.scale, d.causal, d.window_size, d.device = sc, causal, -1, 0
d.stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
d.q, d.k, d.v, d.out, d.dout, d.lse = q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), do.data_ptr(), lse.data_ptr()
d.dq, d.dk, d.dv = dq.data_ptr(), dk.data_ptr(), dv.data_ptr()
n = int(lib.aule_attention_backward_workspace_size(ctypes.byref(d)))
ws = torch.empty(n, device="cuda", dtype=torch.uint8); d.workspace, d.workspace_bytes = ws.data_ptr(), n
stamps = torch.zeros(8 * 384, device="cuda", dtype=torch.int64)
fn = lib.aule_hip_debug_backward_timeline
fn.restype = ctypes.c_int32; fn.argtypes = [ctypes.POINTER(_capi.AttnBwdDesc), ctypes.c_void_p]
for _ in range(10):
    rc = fn(ctypes.byref(d), ctypes.c_void_p(stamps.data_ptr()))
torch.cuda.synchronize()
t = stamps.cpu().tolist()
print(f"rc {rc} causal={causal} B{B} Hq{Hq} Hkv{Hkv} S{S}")
for w in range(4):
    n, a, b, wt = t[4 * w], t[4 * w + 1], t[4 * w + 2], t[4 * w + 3]
    if n:
        print(f"  wave {w}: {n} iterations, phase 1 {(a - wt) / n:7.0f}  boundary wait + barrier {wt / n:7.0f}  phase 2 (+ lgkmcnt wait) {b / n:7.0f}  = {(a + b) / n:7.0f} cycles per iteration ({D // 4} MFMAs = {D * 8})")
