#!/usr/bin/env python3
"""tools/ceiling_control.py -- an INDEPENDENT control for "what does this chip sustain on random data" (VERDICT r4 item 3): the vendor GEMM
(torch.matmul -> hipBLASLt, bf16 8192^3) on N(0,1) operands and on zeros, next to this build's C2 forward, under the same sampler (hwmon
power1_input / power1_cap / freq1_input every 10 ms, means after the first 500 ms), 3 s each, one session.  tools/probe_mfma_power.hip's two
MFMA shapes are run by the session script under their own sampler (same files).  Also dumps which 32-bit words of the gpu_metrics blob move
during a leg (throttler residency accumulators live there on MI300-class parts: format 1.6+ has accumulation_counter, prochot / ppt /
socket_thm / vr_thm / hbm_thm residency at byte offsets 32 .. 52 -- printed with those labels, to be read as tentative)."""
import glob, os, struct, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "aule-attention_amd"))
import torch

def cards():
    out = []
    for c in sorted(glob.glob("/sys/class/drm/card[0-9]*")):
        if "-" in os.path.basename(c):
            continue
        for h in glob.glob(c + "/device/hwmon/hwmon*"):
            if os.path.exists(h + "/power1_input") or os.path.exists(h + "/power1_average"):
                out.append((os.path.basename(c), c + "/device", h))
    return out

def rd(p):
    try:
        with open(p) as f:
            return float(f.read().split()[0])
    except Exception:
        return float("nan")

class Sampler:
    def __init__(self, cs):
        self.cs, self.rows, self.run = cs, [], False
    def start(self):
        self.rows, self.run = [], True
        self.t0 = time.perf_counter()
        def loop():
            while self.run:
                r = [(time.perf_counter() - self.t0) * 1e3]
                for _, dev, h in self.cs:
                    p = rd(h + "/power1_input") if os.path.exists(h + "/power1_input") else rd(h + "/power1_average")
                    r += [p / 1e6, rd(h + "/power1_cap") / 1e6, rd(h + "/freq1_input") / 1e6]
                self.rows.append(r)
                time.sleep(0.01)
        self.th = threading.Thread(target=loop, daemon=True); self.th.start()
    def stop(self):
        self.run = False; self.th.join()
        rows = [r for r in self.rows if r[0] >= 500.0] or self.rows
        out = []
        for i, (name, _, _) in enumerate(self.cs):
            col = lambda k: [r[1 + 3 * i + k] for r in rows]
            m = lambda x: sum(x) / max(1, len(x))
            out.append((name, m(col(0)), max(col(0)), m(col(1)), m(col(2))))
        return out

def metrics_blob(dev):
    try:
        with open(dev + "/gpu_metrics", "rb") as f:
            return f.read()
    except Exception:
        return b""

LABELS = {32: "accumulation_counter?", 36: "prochot_residency_acc?", 40: "ppt_residency_acc?", 44: "socket_thm_residency_acc?", 48: "vr_thm_residency_acc?", 52: "hbm_thm_residency_acc?"}

def leg(name, fn, flops, cs, seconds=3.0):
    fn(); torch.cuda.synchronize()
    before = [metrics_blob(dev) for _, dev, _ in cs]
    s = Sampler(cs); s.start()
    n, t0 = 0, time.perf_counter()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    while time.perf_counter() - t0 < seconds:
        for _ in range(20):
            fn()
        n += 20
        torch.cuda.synchronize()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    st = s.stop()
    after = [metrics_blob(dev) for _, dev, _ in cs]
    tf = flops * n / (ms * 1e-3) / 1e12
    busiest = max(st, key=lambda x: x[1])
    print("leg %-34s %8.1f us/launch  %7.1f TFLOP/s = %.3f of 2516.6 | %s: power mean %.1f W (max %.1f), cap %.0f W, sclk %.0f MHz" % (
        name, ms * 1e3 / n, tf, tf / 2516.6, busiest[0], busiest[1], busiest[2], busiest[3], busiest[4]), flush=True)
    for (cname, _, _), b, a in zip(cs, before, after):
        if cname != busiest[0] or len(b) < 64 or len(a) != len(b):
            continue
        moved = []
        for off in range(4, min(len(b), 160) - 3, 4):
            x, y = struct.unpack_from("<I", b, off)[0], struct.unpack_from("<I", a, off)[0]
            if x != y and off >= 32:
                moved.append("%d:%+d%s" % (off, (y - x) & 0xffffffff, (" " + LABELS[off]) if off in LABELS else ""))
        print("    gpu_metrics %s (size %d, format %d.%d) words that moved: %s" % (cname, b[0] | (b[1] << 8), b[2], b[3], ", ".join(moved[:24])))
    return tf

def main():
    import aule
    cs = cards()
    print("# cards with hwmon power:", [c[0] for c in cs])
    for name, dev, h in cs[:1]:
        print("# %s hwmon files: %s" % (name, " ".join(sorted(os.listdir(h)))))
        for f in ("power1_cap_max", "power1_cap_min", "power1_cap_default", "power1_label", "pp_power_profile_mode"):
            p = (h if f.startswith("power1") else dev) + "/" + f
            if os.path.exists(p):
                print("#   %s: %s" % (f, open(p).read().strip().replace("\n", " | ")[:300]))
    dev = torch.device("cuda")
    g = torch.Generator(device=dev).manual_seed(1)
    N = 8192
    a = torch.randn(N, N, device=dev, dtype=torch.bfloat16, generator=g); b = torch.randn(N, N, device=dev, dtype=torch.bfloat16, generator=g)
    c = torch.empty(N, N, device=dev, dtype=torch.bfloat16)
    z = torch.zeros(N, N, device=dev, dtype=torch.bfloat16)
    leg("torch.matmul bf16 8192^3 N(0,1)", lambda: torch.matmul(a, b, out=c), 2.0 * N ** 3, cs)
    leg("torch.matmul bf16 8192^3 zeros", lambda: torch.matmul(z, z, out=c), 2.0 * N ** 3, cs)
    q, k, v = (torch.randn(4, 32, 4096, 128, device=dev, dtype=torch.bfloat16, generator=g) for _ in range(3))
    fl = 4.0 * 4 * 32 * 128 * (4096 * 4097 // 2)
    def fwd():
        with torch.no_grad():
            aule.flash_attention(q, k, v, causal=True)
    leg("C2 forward N(0,1) (this build)", fwd, fl, cs)
    qz = torch.zeros_like(q)
    def fwdz():
        with torch.no_grad():
            aule.flash_attention(qz, qz, qz, causal=True)
    leg("C2 forward zeros (this build)", fwdz, fl, cs)

if __name__ == "__main__":
    main()
