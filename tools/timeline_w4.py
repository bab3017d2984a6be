#!/usr/bin/env python3
"""Cycle timeline of workgroup 0 of the one-wave-per-SIMD forward (fa_fwd_w4_gfx950.hip, timeline build behind
aule_hip_debug_forward_timeline with AULE_TL=w4; debug library: cd aule-attention_amd/csrc && make dbg).  Every stamp is
(tag << 56) | s_memtime.  Tags: 0x10/0x11 plain step (parity), 0x20+ generic step (PAR + 2 QK + 4 SM), 0x08 idle step,
0x60 / 0x62-3 / 0x64-5 step 0 / step in front of the last tile / last tile in the embedded-request form,
0x18 phase 1 done, 0x19 phase 2 done (then the loop tail), 0x30 prologue, 0x31 S_0 done (wait for the part's first tiles,
barrier, K_0 reads, bare QK^T), 0x33 P_0[A] done (references, softmax of S_0[A], K_1 reads), 0x40 epilogue, 0x42 block A stored,
0x41 block B stored, 0x50 end of the stream.

    python tools/timeline_w4.py [causal] [B] [H] [S] [waves...]
W4_TL_D=64 (with a debug library built with -DW4_TL_D64) and W4_TL_HKV=n: the D = 64 instance, grouped heads.
"""
import ctypes, math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "aule-attention_amd"))
os.environ.setdefault("AULE_LIBRARY_PATH", os.path.join(ROOT, "build", "variants", "libaule_dbg.so"))
os.environ["AULE_TL"] = "w4"
import torch
from aule import _capi

causal = int(sys.argv[1]) if len(sys.argv) > 1 else 1
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4
H = int(sys.argv[3]) if len(sys.argv) > 3 else 32
S = int(sys.argv[4]) if len(sys.argv) > 4 else 4096
waves = [int(x) for x in sys.argv[5:]] or [0, 3]
D, NW, NMAX = int(os.environ.get("W4_TL_D", "128")), 4, 2048
HKV = int(os.environ.get("W4_TL_HKV", "0")) or H
lib = _capi.get_lib()
lib.aule_hip_debug_forward_timeline.restype = ctypes.c_int32
lib.aule_hip_debug_forward_timeline.argtypes = [ctypes.POINTER(_capi.AttnDesc), ctypes.c_void_p]
q = torch.randn(B, H, S, D, device="cuda", dtype=torch.bfloat16)
k, v = (torch.randn(B, HKV, S, D, device="cuda", dtype=torch.bfloat16) for _ in range(2))
out = torch.empty_like(q)
st = torch.zeros(NW * NMAX, device="cuda", dtype=torch.int64)
d = _capi.AttnDesc()
d.struct_size = ctypes.sizeof(_capi.AttnDesc)
d.dtype = 2
d.batch, d.heads_q, d.heads_kv, d.seq_q, d.seq_k, d.head_dim = B, H, HKV, S, S, D
d.scale = 1 / math.sqrt(D)
d.causal = causal
d.window_size = -1
d.device = 0
d.stream = None
d.q, d.k, d.v, d.out, d.lse = q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), None
for _ in range(20):
    st.zero_()
    rc = lib.aule_hip_debug_forward_timeline(ctypes.byref(d), ctypes.c_void_p(st.data_ptr()))
torch.cuda.synchronize()
print("rc", rc, f"causal={causal} B{B} H{H} S{S}")
t = st.cpu().view(NW, NMAX).tolist()
MASK = (1 << 56) - 1
rows = [[((x >> 56) & 0xff, x & MASK) for x in t[w] if x != 0] for w in range(NW)]
t0 = min(r[0][1] for r in rows if r)
tend = max(r[-1][1] for r in rows if r)
print(f"workgroup 0: {tend - t0} cycles from first to last stamp")
for w in waves:
    r = rows[w]
    if os.environ.get("RAW"):   # tag:cycles-to-next-stamp of the first stamps of the wave
        print(f"--- wave {w} raw: " + " ".join(f"{a:#x}:{r[i + 1][1] - b}" for i, (a, b) in enumerate(r[:int(os.environ['RAW'])]) if i + 1 < len(r)))
    print(f"--- wave {w}: {len(r)} stamps, first at +{r[0][1] - t0}")
    i = 0
    part = 0
    acc = {}   # kind -> [n, phase1, phase2, tail]
    def flush():
        for kind, (n, a, b, c_) in sorted(acc.items()):
            print(f"      {kind:10s} x{n:3d}: phase 1 {a / n:7.0f}  phase 2 {b / n:7.0f}  waits+barrier {c_ / n:6.0f}  = {(a + b + c_) / n:7.0f} per step")
        acc.clear()
    while i < len(r):
        tag, tm = r[i]
        nxt = r[i + 1][1] if i + 1 < len(r) else tm
        if tag == 0x30:
            flush()
            seq = [tm]
            j = i + 1
            while j < len(r) and r[j][0] in (0x31, 0x32, 0x33, 0x34, 0x35, 0x36, 0x37):
                seq.append(r[j][1]); j += 1
            end = r[j][1] if j < len(r) else seq[-1]
            d_ = [seq[k + 1] - seq[k] for k in range(len(seq) - 1)] + [end - seq[-1]]
            if len(seq) >= 7:   # the seam form (round 4): 0x30 wait + barrier + K_0 reads + K_3 request | 0x34 QK^T half A + pack A | 0x35 slab A out | 0x36 half B + pack B | 0x37 slab B out | 0x31 refs .. | 0x33
                print(f"   part {part} prologue (seam) at +{tm - t0}: wait + barrier + K_0 read {d_[0]}, QK^T A + pack A {d_[1]}, slab A out {d_[2]}, QK^T B + pack B {d_[3]}, slab B out {d_[4]}, refs + P_0[A] + K_1 read {d_[5]}, to step 0 {d_[6] if len(d_) > 6 else -1}")
            else:
                print(f"   part {part} prologue at +{tm - t0}: wait + barrier + K_0 read + S_0 {d_[0] if len(d_) > 0 else -1}, refs + P_0[A] + K_1 read {d_[1] if len(d_) > 1 else -1}, to step 0 {d_[2] if len(d_) > 2 else -1}")
            part += 1
            i = j
            continue
        if tag in (0x10, 0x11) or 0x20 <= tag < 0x30 or 0x60 <= tag < 0x68:
            # 0x60 step 0 / 0x62+PAR step in front of the wave's last tile / 0x64+PAR the last tile: embedded-request bodies (round 4)
            kind = "plain" if tag < 0x20 else (f"gen{tag:#x}" if tag < 0x60 else {0x60: "first", 0x62: "prediag", 0x63: "prediag", 0x64: "diag", 0x65: "diag"}.get(tag, f"fast{tag:#x}"))
            j = i + 1
            p1 = p2 = None
            extra = []
            while j < len(r) and r[j][0] in (0x18, 0x19, 0x1a, 0x1b):
                if r[j][0] == 0x18: p1 = r[j][1]
                elif r[j][0] == 0x19: p2 = r[j][1]
                else: extra.append(r[j])
                j += 1
            if extra and os.environ.get("ALL"):
                print(f"      {kind} at +{tm - t0}: p1 {p1 - tm} p2 {p2 - p1} " + " ".join(f"{a:#x}:+{b - p2}" for a, b in extra) + f" next +{(r[j][1] if j < len(r) else 0) - p2}")
            end = r[j][1] if j < len(r) else tm
            if p1 is not None and p2 is not None:
                a = acc.setdefault(kind, [0, 0, 0, 0])
                a[0] += 1; a[1] += p1 - tm; a[2] += p2 - p1; a[3] += end - p2
            i = j
            continue
        if tag == 0x08:
            a = acc.setdefault("idle", [0, 0, 0, 0])
            a[0] += 1; a[3] += nxt - tm
            i += 1
            continue
        if tag == 0x40:
            flush()
            seq = [tm]
            j = i + 1
            while j < len(r) and r[j][0] in (0x42, 0x41):
                seq.append(r[j][1]); j += 1
            end = r[j][1] if j < len(r) else seq[-1]
            print(f"   epilogue at +{tm - t0}: block A (pack, slab, stores) {seq[1] - seq[0] if len(seq) > 1 else -1}, block B {seq[2] - seq[1] if len(seq) > 2 else -1}, to next {end - seq[-1]}")
            i = j
            continue
        if tag == 0x50:
            flush()
            print(f"   end of stream at +{tm - t0}")
        i += 1
    flush()
