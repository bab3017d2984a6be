import math, os, sys, torch
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "aule-attention_amd"))
import aule
def t(f, n=20):
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for D in (64, 128):
    for dt in (torch.bfloat16, torch.float16):
        for S, W in ((4096, 256), (4096, 1024)):
            q = torch.randn(4, 32, S, D, device="cuda", dtype=dt); k = torch.randn_like(q); v = torch.randn_like(q)
            us = t(lambda: aule.flash_attention(q, k, v, causal=True, window_size=W))
            # same, with the first 256 rows cut off the timing: queries 256.. only (bottom-right: no row with a single key)
            q2 = q[:, :, 256:].contiguous()
            us2 = t(lambda: aule.flash_attention(q2, k, v, causal="bottom-right", window_size=W))
            print(f"D{D} {str(dt)[6:]} S{S} W{W}: {us:8.1f} us   rows 256.. only (bottom-right): {us2:8.1f} us", flush=True)
