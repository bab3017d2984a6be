#!/usr/bin/env python3
"""Cycle timeline of workgroup 0 of the persistent tile-stream forward (fa_fwd_ps_gfx950.hip, timeline build behind
aule_hip_debug_forward_timeline with AULE_TL=ps).  Every stamp is (tag << 56) | s_memtime; per tile step the tags are
base + {1 V-phase start, 2 staging done, 3 Q request / epilogue done, 4 softmax done, 5 barrier passed (M-phase start),
6 PV done (seam: Q taken), 7 M-phase done}, base = 8*MODE + 32*FIRST + 64*SEAM.

    AULE_TL=ps python tools/timeline_ps.py [causal] [B] [H] [S] [waves...]
"""
import ctypes, math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "aule-attention_amd"))
# the timeline hooks live in the debug library only (cd aule-attention_amd/csrc && make dbg)
os.environ.setdefault("AULE_LIBRARY_PATH", os.path.join(ROOT, "build", "variants", "libaule_dbg.so"))
import torch
from aule import _capi

causal = int(sys.argv[1]) if len(sys.argv) > 1 else 1
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
H = int(sys.argv[3]) if len(sys.argv) > 3 else 16
S = int(sys.argv[4]) if len(sys.argv) > 4 else 1024
waves = [int(x) for x in sys.argv[5:]] or [0, 3, 4, 7]
D, NW, NMAX = 128, 8, 2048
os.environ.setdefault("AULE_TL", "ps")
lib = _capi.get_lib()
lib.aule_hip_debug_forward_timeline.restype = ctypes.c_int32
lib.aule_hip_debug_forward_timeline.argtypes = [ctypes.POINTER(_capi.AttnDesc), ctypes.c_void_p]
q, k, v = (torch.randn(B, H, S, D, device="cuda", dtype=torch.bfloat16) for _ in range(3))
out = torch.empty_like(q)
st = torch.zeros(NW * NMAX, device="cuda", dtype=torch.int64)
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
for _ in range(3):
    st.zero_()
    rc = lib.aule_hip_debug_forward_timeline(ctypes.byref(d), ctypes.c_void_p(st.data_ptr()))
    torch.cuda.synchronize()
print("rc", rc, f"causal={causal} B{B} H{H} S{S}")
t = st.cpu().view(NW, NMAX).tolist()
MASK = (1 << 56) - 1
rows = []
for w in range(NW):
    rows.append([((x >> 56) & 0xff, x & MASK) for x in t[w] if x != 0])
t0 = min(r[0][1] for r in rows if r)
tend = max(r[-1][1] for r in rows if r)
print(f"workgroup 0: {tend - t0} cycles from first to last stamp")
for w in waves:
    r = rows[w]
    print(f"--- wave {w}: {len(r)} stamps, first at +{r[0][1] - t0}")
    steps = []          # [base, {code: time}, [epilogue stamps]]
    cur = None
    for tag, tm in r:
        if 0xd0 <= tag < 0xe0:
            if cur is not None:
                cur[2].append((tag, tm))
            continue
        if tag >= 0xe0:
            steps.append((tag, {0: tm}, []))
            cur = None
            continue
        base, code = tag & ~7, tag & 7
        if cur is None or cur[0] != base or code in cur[1] or code < max(cur[1]):
            cur = (base, {}, [])
            steps.append(cur)
        cur[1][code] = tm
    n = 0
    for i, (base, seg, epi) in enumerate(steps):
        nxt = min(steps[i + 1][1].values()) if i + 1 < len(steps) else None
        if base >= 0xe0:
            print(f"   [{base:#x}] at +{seg[0] - t0}  (+{(nxt - seg[0]) if nxt else 0} to next)")
            continue
        mode, first, seam = (base >> 3) & 3, (base >> 5) & 1, (base >> 6) & 1
        g = lambda a, b: (seg[b] - seg[a]) if a in seg and b in seg else -1
        end = max(seg.values())
        kind = ("FIRST " if first else "") + ("SEAM " if seam else "") + f"M{mode}"
        if first or seam or mode != 2 or n < 3 or os.environ.get("ALL"):
            print(f"   step {n:3d} {kind:12s} at +{seg.get(1, end) - t0:7d}: stage {g(1,2):5d} q/epi {g(2,3):5d} softmax {g(3,4):5d} bar {g(4,5):5d} | "
                  f"PV/takeQ {g(5,6):5d} QK {g(6,7) if 6 in seg else g(5,7):5d} | to next {(nxt - end) if nxt else -1:5d}  total {(nxt - seg[1]) if nxt and 1 in seg else -1}")
            if epi:
                prev = seg.get(2, epi[0][1])
                out = []
                for tag, tm in epi:
                    out.append(f"{tag:#x}:+{tm - prev}")
                    prev = tm
                print("        epilogue: " + " ".join(out))
        n += 1
