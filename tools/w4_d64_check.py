import sys
sys.path.insert(0, "tools")
import fwd_check as pc
ok = True
for c in [("fp16", 1, 32, 1, 4096, 4096, 64, False), ("fp16", 8, 32, 32, 2048, 2048, 64, True), ("bf16", 8, 32, 32, 2048, 2048, 64, True),
          ("bf16", 1, 4, 4, 512, 512, 64, False), ("fp16", 2, 4, 4, 600, 600, 64, True), ("bf16", 2, 8, 2, 777, 1300, 64, False),
          ("fp16", 1, 2, 2, 256, 256, 64, True), ("bf16", 1, 8, 8, 512, 1024, 64, "bottom-right")]:
    ok &= pc.check(*c, want_route=None)
ok &= pc.check("fp16", 2, 8, 2, 1024, 1024, 64, False, mag=4.0, want_route=None)
ok &= pc.check("fp16", 16, 16, 16, 512, 33024, 64, False, qzero=True, want_route=None)
print("D64 ALL OK" if ok else "D64 SOME FAILED")
pc.bench_shape("fp16", 1, 32, 1, 16384, 64, False, warm=30, iters=30)
pc.bench_shape("bf16", 8, 32, 32, 2048, 64, True, warm=100, iters=60)
pc.bench_shape("fp16", 4, 32, 8, 4096, 64, True, warm=60, iters=40)
