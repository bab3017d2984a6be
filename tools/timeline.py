#!/usr/bin/env python3
"""Per-phase cycle timeline of workgroup 0 of the ping-pong forward kernel (debug build hook
aule_hip_debug_forward_timeline): prints, per wave, the V-phase / barrier / M-phase / barrier
durations in shader cycles for the first tiles."""
import ctypes
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "aule-attention_amd"))
# the timeline hooks live in the debug library only (cd aule-attention_amd/csrc && make dbg)
os.environ.setdefault("AULE_LIBRARY_PATH", os.path.join(ROOT, "build", "variants", "libaule_dbg.so"))
import torch
from aule import _capi

causal = int(sys.argv[1]) if len(sys.argv) > 1 else 0
S = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
NT = int(sys.argv[3]) if len(sys.argv) > 3 else 10
B, H, D = 2, 8, 128
lib = _capi.get_lib()
lib.aule_hip_debug_forward_timeline.restype = ctypes.c_int32
lib.aule_hip_debug_forward_timeline.argtypes = [ctypes.POINTER(_capi.AttnDesc), ctypes.c_void_p]
q, k, v = (torch.randn(B, H, S, D, device="cuda", dtype=torch.bfloat16) for _ in range(3))
out = torch.empty_like(q)
st = torch.zeros(8 * 256, device="cuda", dtype=torch.int64)
d = _capi.AttnDesc()
d.struct_size = ctypes.sizeof(_capi.AttnDesc)
d.dtype = 2
d.batch, d.heads_q, d.heads_kv, d.seq_q, d.seq_k, d.head_dim = B, H, H, S, S, D
d.scale = 1 / math.sqrt(D)
d.causal = causal
d.window_size = -1
d.device = 0
d.stream = None
d.q, d.k, d.v, d.out, d.lse = q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), None
for _ in range(2):
    rc = lib.aule_hip_debug_forward_timeline(ctypes.byref(d), ctypes.c_void_p(st.data_ptr()))
torch.cuda.synchronize()
print("rc", rc)
t = st.cpu().view(8, 256)
if os.environ.get("RAW"):
    for w in (0, 7):
        row = [int(x) for x in t[w] if int(x) > 0]
        print("wave", w, "deltas:", [b - a for a, b in zip(row, row[1:])])
    sys.exit(0)
t0 = int(t[:, 0].min())
for w in range(8):
    row = t[w]
    print(f"wave {w}: start+{int(row[0]) - t0}")
    segs = []
    # stamps per tile: V0, after-rowmax (classic softmax only), V-end, M0, after-PV, after-QK, M-end
    NS = 7 if os.environ.get("AULE_HIP_FWD_SOFTMAX", "raw")[0] == "c" else 6
    for j in range(0, NT):
        if NS == 7:
            v0, v1, v2, m0, m1, m2, m3 = (int(row[NS * j + i]) for i in range(NS))
        else:
            v0, v2, m0, m1, m2, m3 = (int(row[NS * j + i]) for i in range(NS))
            v1 = v0
        nxt = int(row[NS * j + NS])
        segs.append(f"max{v1 - v0:5d} exp{v2 - v1:5d} bar{m0 - v2:5d} PV{m1 - m0:5d} QK{m2 - m1:5d} wr{m3 - m2:5d} bar{nxt - m3:5d}")
    for k in range(0, len(segs), 4):
        print("   " + " | ".join(segs[k:k + 4]))
