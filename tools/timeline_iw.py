#!/usr/bin/env python3
"""Per-segment cycle timeline of workgroup 0 of the in-wave ping-pong forward kernel
(AULE_HIP_FWD_KERNEL=iw, debug hook aule_hip_debug_forward_timeline): per wave and tile step the X segment,
Y segment, staging writes and barrier wait in shader cycles."""
import ctypes, math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "aule-attention_amd"))
os.environ.setdefault("AULE_HIP_FWD_KERNEL", "iw")
import torch
from aule import _capi

causal = int(sys.argv[1]) if len(sys.argv) > 1 else 0
S = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
NT = int(sys.argv[3]) if len(sys.argv) > 3 else 12
B, H, D = 2, 8, 128
lib = _capi.get_lib()
lib.aule_hip_debug_forward_timeline.restype = ctypes.c_int32
lib.aule_hip_debug_forward_timeline.argtypes = [ctypes.POINTER(_capi.AttnDesc), ctypes.c_void_p]
q, k, v = (torch.randn(B, H, S, D, device="cuda", dtype=torch.bfloat16) for _ in range(3))
out = torch.empty_like(q)
NW = 8 if os.environ["AULE_HIP_FWD_KERNEL"] == "iw1" else 4
st = torch.zeros(8 * 512, device="cuda", dtype=torch.int64)
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
t = st.cpu().view(8, 512)
t0 = int(t[:, 0].min())
for w in range(NW):
    row = [int(x) for x in t[w]]
    print(f"wave {w}: start+{row[0] - t0}")
    segs = []
    for j in range(NT):   # 4 stamps per tile step: start, after X, after Y, after barrier
        a, b, c, e = row[4 * j: 4 * j + 4]
        nxt = row[4 * j + 4]
        segs.append(f"X{b - a:5d} Y{c - b:5d} wr+bar{e - c:5d} gap{nxt - e:4d}")
    for i in range(0, len(segs), 4):
        print("   " + " | ".join(segs[i:i + 4]))
    tot = row[4 * NT] - row[4]
    print(f"   steps 1..{NT-1}: {tot / (NT - 1):.0f} cycles per tile")
