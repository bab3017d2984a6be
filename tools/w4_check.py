#!/usr/bin/env python3
"""Triage and timing of the one-wave-per-SIMD forward (fa_fwd_w4_gfx950.hip, route 8).

    python tools/w4_check.py check [quick]   parity vs an fp64 reference on the GPU, shapes that exercise every step variant
    python tools/w4_check.py bench [tag]     per-launch times of the headline shapes for the kernel the environment selects
                                             (AULE_HIP_FWD_KERNEL=pp: the two-waves-per-SIMD ping-pong kernel)
The judge and the timing loop are those of tools/fwd_check.py.
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import fwd_check as pc


def run_checks(quick):
    ok = True
    C = [
        # dtype, B, Hq, Hkv, Sq, Sk, D, causal
        ("bf16", 1, 2, 2, 256, 256, 128, True),        # one block, 4 tiles: waves see 1, 2, 3, 4 tiles
        ("bf16", 1, 2, 2, 256, 256, 128, False),       # no mask at all
        ("bf16", 1, 2, 2, 300, 300, 128, True),        # pair, ragged (5 tiles: odd part + padding step)
        ("bf16", 1, 4, 2, 512, 512, 128, True),
        ("bf16", 1, 2, 2, 200, 333, 128, False),       # ragged last tile
        ("bf16", 1, 2, 2, 777, 260, 128, False),       # 5 tiles (odd), 4 blocks, padding rows
        ("bf16", 2, 4, 1, 1024, 1024, 128, True),
        ("bf16", 1, 3, 3, 1280, 1280, 128, True),      # 5 blocks: the middle one unpaired
        ("bf16", 1, 8, 8, 1024, 512, 128, True),       # Sq > Sk, top-left
        ("bf16", 1, 8, 8, 512, 1024, 128, "bottom-right"),
        ("fp16", 2, 8, 8, 1024, 1024, 128, True),
        ("bf16", 1, 64, 64, 1024, 1024, 128, False),
        ("bf16", 4, 32, 8, 2048, 2048, 128, True),     # C3 forward (GQA)
    ]
    if not quick:
        C += [
            ("bf16", 4, 32, 32, 2048, 2048, 128, True),    # 512 items: two per workgroup
            ("bf16", 4, 32, 32, 4096, 4096, 128, True),    # C2: four items = eight parts per workgroup
            ("bf16", 2, 40, 8, 1280, 1280, 128, False),    # 400 items
            ("bf16", 4, 32, 32, 4096, 4096, 128, False),
            ("bf16", 4, 32, 8, 1000, 3000, 128, "bottom-right"),
            ("fp16", 4, 16, 16, 1111, 1111, 128, True),
        ]
    for c in C:
        ok &= pc.check(*c, want_route=8)
    ok &= pc.check("bf16", 1, 2, 2, 512, 512, 128, True, mag=6.0, want_route=8)      # fixed-reference range fails -> exact-maximum pass
    ok &= pc.check("bf16", 1, 2, 2, 512, 512, 128, False, mag=12.0, want_route=8)
    ok &= pc.check("fp16", 1, 2, 2, 512, 512, 128, True, mag=6.0, want_route=8)      # fp16: weights overflow -> second stream
    if not quick:
        ok &= pc.check("bf16", 4, 32, 32, 2048, 2048, 128, True, mag=5.0, want_route=8)  # ... in the middle of long part lists
    ok &= pc.check("bf16", 2, 4, 4, 1024, 1024, 128, False, scale=0.3, want_route=8)
    ok &= pc.check("bf16", 2, 4, 4, 1024, 1024, 128, True, scale=-0.1, want_route=1)     # negative scale: the ping-pong kernel
    print("ALL OK" if ok else "SOME FAILED", flush=True)
    return ok


def run_bench(tag):
    print(f"bench [{tag}] kernel={os.environ.get('AULE_HIP_FWD_KERNEL', '(default)')} route C2={pc.route('bf16', 4, 32, 32, 4096, 4096, 128, True)}", flush=True)
    pc.bench_shape("bf16", 4, 32, 32, 4096, 128, True)
    pc.bench_shape("bf16", 4, 32, 32, 4096, 128, False)
    pc.bench_shape("bf16", 4, 32, 8, 2048, 128, True)
    pc.bench_shape("bf16", 8, 32, 32, 8192, 128, True, warm=20, iters=20)
    pc.bench_shape("bf16", 16, 16, 16, 1024, 128, True)


if __name__ == "__main__":
    mode = sys.argv[1] if len(sys.argv) > 1 else "check"
    if mode == "check":
        sys.exit(0 if run_checks(len(sys.argv) > 2 and sys.argv[2] == "quick") else 1)
    elif mode == "bench":
        run_bench(sys.argv[2] if len(sys.argv) > 2 else "")
