#!/usr/bin/env python3
"""bench.py -- throughput of the FlashAttention hot path on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config c2|c3|c4|c5] [--mode fwd|fwdbwd] [--condition-ms MS]
    (--config defaults to c2, the single-GPU headline, on one GPU and to c4 -- BASELINE configs[3]: B=64 over 8 GPUs, 8 per GPU, S=8192 -- with --gpus N > 1)

One "step" = one pass of the hot path (aule.flash_attention -> libaule.so -> gfx950
kernels) over one batch of synthetic [B,H,S,D] tensors already resident in HBM.
Default workload = BASELINE.json configs[1]:  B=4 H=32 S=4096 D=128 bf16 causal MHA, fwd.
With N > 1 (launched by torch.distributed.run, one rank per GPU, RCCL) every rank runs
configs[3]'s per-GPU batch (B=8 H=32 S=8192) on its own shard (weak scaling, no data-path collective); the
optional output all-gather over xGMI is timed separately and reported under "gather".

Rank 0 prints ONE JSON line.  FLOP convention: fwd = 4*B*Hq*D*P with
P = sum_i min(i+1, Sk) for the top-left causal mask, bwd = 2.5*fwd.

Protocol.  The forward leg runs with autograd enabled, so the kernel stores the log-sum-exp like the reference's forward
always does (python/aule/triton_flash_amd.py:410-432).  `value` is the contract's measurement and nothing else: W untimed
warm-up steps, then EXACTLY K timed steps between barrier + synchronize pairs, taken first, on the chip as the process finds
it.  MI355X clocks to its power budget: from an idle chip the first two launches of this kernel run at boost clock, launches
3-8 collapse to ~1.45x the steady time while the power controller overshoots, and it takes ~60 launches (35-40 ms) to settle
(profiles/r2_dvfs_trace.txt; same curve on every box) -- a 5 + 20 step measurement right after process start sits inside
that transient.  The number a long-running job sees is therefore measured too and reported SEPARATELY as "steady_state":
the same step repeated for --condition-ms of device time (default 250 ms), then W warm-up + K timed steps again.  (Round 2
printed the conditioned figure as `value` and the plain one as "cold_start"; the two have swapped places.)  Every timed step
has its own HIP event pair: mean, median, min and max per launch are in "roofline".
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "aule-attention_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

# gfx950 dense peaks (MI355X_MICROARCH.md: 256 CU x 4 SIMD x 1024 FLOP/clk x 2.4 GHz; "~2.5 PF dense")
PEAK_TFLOPS = {"bf16": 2516.6, "fp16": 2516.6, "fp32": 157.3}

CONFIGS = {
    # name: (B per GPU, Hq, Hkv, Sq, Sk, D, dtype, causal, default mode)
    "c2": (4, 32, 32, 4096, 4096, 128, "bf16", True, "fwd"),      # configs[1] -- the headline
    "c3": (4, 32, 8, 2048, 2048, 128, "bf16", True, "fwdbwd"),    # configs[2] (B=4 assumed)
    "c4": (8, 32, 32, 8192, 8192, 128, "bf16", True, "fwd"),      # configs[3]: B=64 over 8 GPUs
    "c5": (1, 32, 1, 16384, 16384, 64, "fp16", False, "fwd"),     # configs[4]
    # SURVEY 8d note: the genuinely HBM-bound cross-attention points that bracket the regime configs[4] is labelled with
    # c5b / c5c: K+V is 4 MB, resident in L2 / the Infinity Cache across steps -- their "hbm" roofline entry is the
    # algorithmic byte rate of a launch-bound call, not DRAM traffic (tools/bench_decode_ws.py has the HBM rates)
    "c5b": (1, 32, 1, 1, 16384, 64, "fp16", False, "fwd"),        # decode-like: AI = 32 FLOP/B
    "c5c": (1, 32, 1, 64, 16384, 64, "fp16", False, "fwd"),       # AI ~ 1800 FLOP/B
}
HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E ~8 TB/s
RIDGE = 315.0           # FLOP/B where the bf16/fp16 MFMA roof meets the HBM roof


def causal_pairs(Sq, Sk):
    n = min(Sq, Sk)
    return n * (n + 1) // 2 + max(0, Sq - Sk) * Sk


def fwd_flops(B, Hq, Sq, Sk, D, causal):
    P = causal_pairs(Sq, Sk) if causal else Sq * Sk
    return 4.0 * B * Hq * D * P


def stored_profile(config, mode):
    """What the LAST rocprofv3 run of this workload recorded (tools/profile.sh -> profiles/hbm_traffic.json): HBM bytes per
    launch from the PMC passes (FETCH_SIZE x2 per the gfx950 note in MI355X_MICROARCH.md, + WRITE_SIZE) and the effective
    clock (GRBM_GUI_ACTIVE / 8 XCDs / kernel time).  STORED figures of a profiled run, not measurements of this one."""
    try:
        with open(os.path.join(ROOT, "profiles", "hbm_traffic.json")) as fh:
            return json.load(fh).get("%s_%s" % (config, mode), {})
    except (OSError, ValueError):
        return {}


def hbm_traffic(config, mode):
    return stored_profile(config, mode).get("bytes_per_launch")


def _cpu_model():
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.lower().startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(budget_s=10.0):
    """The oracle's NumPy restatement of the reference CPU path (python/aule/__init__.py:247-271) on this box's host cores,
    by SURVEY.md 8d's protocol: config C1 exactly (B1 H8 S256 D64 fp32 causal; 3 warm-ups, median of 20) is `value`; one
    larger point that still fits the O(S^2) temporaries (B1 H8 S2048 D64) and a bounded sample of the bench workload itself
    (one batch element, 2 of the 32 heads of C2) ride along as structured fields.  BLAS thread settings are left at their
    defaults and printed; `cores` = CPU time / wall time of the C1 loop (the threads actually busy)."""
    import numpy as np
    import oracle
    rng = np.random.RandomState(0)

    def point(B, H, S, D, warm, reps, budget=None):
        q, k, v = (rng.randn(B, H, S, D).astype(np.float32) for _ in range(3))
        for _ in range(warm):
            oracle.cpu_attention(q, k, v, True)
        ts, c0, w0 = [], time.process_time(), time.perf_counter()
        for _ in range(reps):
            a = time.perf_counter()
            oracle.cpu_attention(q, k, v, True)
            ts.append(time.perf_counter() - a)
            if budget is not None and time.perf_counter() - w0 > budget:
                break
        wall, cpu = time.perf_counter() - w0, time.process_time() - c0
        med = sorted(ts)[len(ts) // 2]
        fl = fwd_flops(B, H, S, S, D, True)
        return {"shape": "B%d H%d S%d D%d fp32 causal" % (B, H, S, D), "reps": len(ts), "warmups": warm, "median_ms": med * 1e3,
                "tflops": fl / med / 1e12, "flops": fl, "threads_busy": round(cpu / wall, 2)}

    c1 = point(1, 8, 256, 64, 3, 20)
    big = point(1, 8, 2048, 64, 1, 5, budget=budget_s / 2)
    c2s = point(1, 2, 4096, 128, 0, 4, budget=budget_s / 2)
    return {
        "value": c1["tflops"],
        "unit": "TFLOP/s",
        "cores": max(1, int(round(c1["threads_busy"]))),
        "host_cores": os.cpu_count(),
        "cpu_model": _cpu_model(),
        "blas_threads": {k: os.environ.get(k, "default") for k in ("OMP_NUM_THREADS", "MKL_NUM_THREADS", "OPENBLAS_NUM_THREADS")},
        "kind": "port",
        "sample": "numpy restatement of _cpu_attention (oracle.cpu_attention) at config C1 exactly: B1 H8 S256 D64 fp32 causal, "
                  "3 warm-ups, median of 20 = %.2f ms (the reference's own code took 40.6 ms on 8 Xeon cores, BASELINE.md)" % c1["median_ms"],
        "c1": c1,
        "b1_h8_s2048_d64": big,
        "c2_sample": c2s,
    }


def workload_string(config, B, mode=None):
    _, Hq, Hkv, Sq, Sk, D, dtype, causal, dmode = CONFIGS[config]
    return "%s: B=%d/GPU Hq=%d Hkv=%d Sq=%d Sk=%d D=%d %s %s %s" % (config, B, Hq, Hkv, Sq, Sk, D, dtype, "causal" if causal else "non-causal", mode or dmode)


def self_spawn(args):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: become `python -m torch.distributed.run --nnodes=1
    --nproc-per-node N --master-addr 127.0.0.1 --master-port <free> bench.py <the same arguments>` -- the command the driver
    uses -- by exec, so that rank 0's JSON line is this process's stdout and its exit code this process's.  The device count is
    checked first: a box with fewer GPUs fails here with that message, not inside a rendezvous."""
    if not args.dry_run:
        import torch
        n = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if n < args.gpus:
            raise SystemExit("bench.py: --gpus %d needs %d devices, this node shows %d" % (args.gpus, args.gpus, n))
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stdout.flush()
    os.execv(sys.executable, cmd)


def dry_run(args):
    """The bench's launch contract without a device: process group (gloo) of --gpus ranks, the rank-count assertion, K timed
    no-op steps between barriers, MAX over ranks, rank 0 prints one line with "dry_run": true and no throughput."""
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world > 1 or "RANK" in os.environ:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
        assert dist.get_world_size() == world
    else:
        dist = None
    if args.gpus != world:
        raise SystemExit("bench.py: --gpus %d but the launcher made %d rank(s)" % (args.gpus, world))
    for _ in range(args.warmup):
        pass
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        pass
    wall = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([wall], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall = float(t[0])
        dist.barrier()
    if rank == 0:
        print(json.dumps({"metric": "attention TFLOPS/GPU (fwd, fwd+bwd) + % MFMA roofline at S=4096,D=128", "value": None,
                          "unit": "TFLOP/s", "n_gpus": world, "world_size": world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": wall * 1e3 / max(1, args.steps), "dry_run": True,
                          "config": {"workload": workload_string(args.config, args.batch or CONFIGS[args.config][0], args.mode)}}), flush=True)
    if dist is not None:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", default=None, choices=sorted(CONFIGS),
                    help="default: c2 (BASELINE configs[1], the headline) on one GPU, c4 (configs[3]: B=64 over 8 GPUs -> 8 per GPU, S=8192) with --gpus N > 1")
    ap.add_argument("--mode", default=None, choices=["fwd", "fwdbwd"])
    ap.add_argument("--batch", type=int, default=None, help="override the per-GPU batch")
    ap.add_argument("--condition-ms", type=float, default=250.0,
                    help="device time spent repeating the step before warm-up (0: none); see the module docstring")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true")
    ap.add_argument("--dry-run", action="store_true",
                    help="launch / rendezvous / timing scaffolding only (gloo, no device, no kernels): what tests/test_bench_spawn.py drives")
    args = ap.parse_args()

    if args.config is None:
        # BASELINE.json: configs[1] is the single-GPU headline, configs[3] the multi-GPU scaling run (B=64 H=32 S=8192 D=128 over 8 GPUs =
        # 8 per GPU; weak scaling keeps that per-GPU batch at N = 2, 4)
        args.config = "c2" if args.gpus == 1 else "c4"
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return self_spawn(args)

    if args.dry_run:
        return dry_run(args)

    import torch
    import aule  # raises AuleError at first use if libaule.so / a HIP device is missing

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no ROCm device visible (there is no CPU fallback)")
    if torch.cuda.device_count() < max(args.gpus, local_rank + 1):
        raise SystemExit("bench.py: --gpus %d needs %d devices, this node shows %d" % (args.gpus, args.gpus, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or "RANK" in os.environ:   # launched by torch.distributed.run (also exercised with 1 rank)
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world,
                                device_id=torch.device("cuda", local_rank))
    if dist is not None:
        assert dist.get_world_size() == world
    if args.gpus != world:
        # (the driver launches `--gpus N` under torch.distributed.run with N ranks; a plain `python bench.py --gpus N` spawns
        # them itself: self_spawn.  What is left is a launcher that made a different number of ranks than --gpus says.)
        raise SystemExit("bench.py: --gpus %d but the launcher made %d rank(s)" % (args.gpus, world))
    n_gpus = world

    B, Hq, Hkv, Sq, Sk, D, dtype, causal, mode = CONFIGS[args.config]
    if args.batch:
        B = args.batch
    mode = args.mode or mode
    tdt = {"bf16": torch.bfloat16, "fp16": torch.float16, "fp32": torch.float32}[dtype]
    dev = torch.device("cuda", local_rank)
    gen = torch.Generator(device=dev).manual_seed(1234 + rank)
    q = torch.randn(B, Hq, Sq, D, device=dev, dtype=tdt, generator=gen)
    k = torch.randn(B, Hkv, Sk, D, device=dev, dtype=tdt, generator=gen)
    v = torch.randn(B, Hkv, Sk, D, device=dev, dtype=tdt, generator=gen)
    do = torch.randn(B, Hq, Sq, D, device=dev, dtype=tdt, generator=gen) if mode == "fwdbwd" else None
    # autograd on in both modes: the forward then stores LSE (what a training forward, and the reference's, does)
    q.requires_grad_(True); k.requires_grad_(True); v.requires_grad_(True)

    def step():
        if mode == "fwd":
            return aule.flash_attention(q, k, v, causal=causal)
        q.grad = k.grad = v.grad = None
        out = aule.flash_attention(q, k, v, causal=causal)
        out.backward(do)
        return out

    def sync_all():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    last_launches = []

    def timed(fn, steps):
        """Barrier + synchronize, K steps, synchronize; wall clock and HIP-event time on the launch stream (torch's
        current stream = the stream the kernels are launched on), MAX over ranks.  Per-step event pairs give the
        per-launch spread (rank-local)."""
        sync_all()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
        t0 = time.perf_counter()
        ev[0].record()
        for i in range(steps):
            fn()
            ev[i + 1].record()
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        dev_ms = ev[0].elapsed_time(ev[steps])
        last_launches[:] = [ev[i].elapsed_time(ev[i + 1]) for i in range(steps)]
        if dist is not None:
            t = torch.tensor([wall, dev_ms], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            wall, dev_ms = float(t[0]), float(t[1])
            dist.barrier()
        return wall, dev_ms

    def condition(fn, ms):
        """Repeat the step for ~ms of device time (power / clock controller into steady state); returns (steps, ms).
        The step is sized from the mean of three calls AFTER two untimed ones: the first call of a new workload pays its
        allocations (round 3 sized the loop from that call alone -- 5 ms instead of 0.6 -- and the "conditioned" extra legs
        then ran 30 ms, not 250, and were read inside the clock transient: profiles/r4_bench_conditioning.txt)."""
        if ms <= 0:
            return 0, 0.0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        fn()
        fn()
        e0.record()
        for _ in range(3):
            fn()
        e1.record()
        torch.cuda.synchronize()
        first = e0.elapsed_time(e1)
        one = max(1e-3, first / 3)
        n = max(1, int(ms / one))
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return n + 3, e0.elapsed_time(e1) + first      # (the two untimed calls are neither counted nor timed)

    # 1. the contract's protocol, first thing, on the chip as the process finds it: W untimed steps, EXACTLY K timed steps -> `value`
    for _ in range(args.warmup):
        step()
    wall, dev_ms = timed(step, args.steps)
    launches = sorted(last_launches)
    # 2. the same protocol again behind a conditioning phase -> "steady_state" (reported beside `value`, never as it)
    cond_steps, cond_ms = condition(step, args.condition_ms)
    for _ in range(args.warmup):
        step()
    steady_wall, steady_dev_ms = timed(step, args.steps)

    f_fwd = fwd_flops(B, Hq, Sq, Sk, D, causal)
    f_step = f_fwd * (3.5 if mode == "fwdbwd" else 1.0)

    def launch_stats():
        n = len(launches)
        return {"kernel_ms_median": launches[n // 2], "kernel_ms_min": launches[0], "kernel_ms_max": launches[-1]}

    ms_per_step = wall * 1e3 / args.steps
    value = f_step * n_gpus * args.steps / wall / 1e12
    kern_ms = dev_ms / args.steps          # HIP events on the launch stream, per step
    achieved = f_step / (kern_ms * 1e-3) / 1e12
    elt = 2 if dtype != "fp32" else 4
    alg_bytes = elt * (2 * B * Hq * Sq * D + 2 * B * Hkv * Sk * D) + 4 * B * Hq * Sq
    hbm_bound = mode == "fwd" and f_step / alg_bytes < RIDGE

    result = {
        "metric": "attention TFLOPS/GPU (fwd, fwd+bwd) + % MFMA roofline at S=4096,D=128",
        "value": value,
        "unit": "TFLOP/s",
        "n_gpus": n_gpus,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": dtype,
        "data": "synthetic N(0,1) q,k,v resident in HBM, torch.Generator(seed 1234+rank)",
        "config": {"workload": workload_string(args.config, B, mode),
            "global_batch": B * n_gpus, "parallelism": "batch-sharded dp%d, no data-path collective" % n_gpus,
            "flop_convention": "4*B*Hq*D*sum_i min(i+1,Sk) (causal), bwd=2.5x fwd",
            "lse": "stored (autograd is on: the forward writes the log-sum-exp like the reference's, triton_flash_amd.py:410-432)"},
        "steady_state": {"value": f_step * n_gpus * args.steps / steady_wall / 1e12, "ms_per_step": steady_wall * 1e3 / args.steps,
                         "frac": f_step / (steady_dev_ms / args.steps * 1e-3) / 1e12 / PEAK_TFLOPS[dtype],
                         "conditioning_ms": cond_ms, "conditioning_steps": cond_steps,
                         "note": "the same W + K protocol AFTER repeating the step for --condition-ms of device time: past the MI355X DVFS "
                                 "transient that follows load onset (~60 launches / 40 ms, profiles/r2_dvfs_trace.txt).  What a long-"
                                 "running job sees; `value` above is the plain protocol (round 2 called that one cold_start)"},
        "world_size": world,
        "per_gpu_tflops": value / n_gpus,
        "roofline": ({"bound": "hbm", "achieved": alg_bytes / (kern_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                      "frac": alg_bytes / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": hbm_traffic(args.config, mode),
                      "kernel_ms": kern_ms, **launch_stats(),
                      "note": "arithmetic intensity %.0f FLOP/B < ridge %.0f: achieved = algorithmic bytes (%d) / HIP-event "
                              "time per step" % (f_step / alg_bytes, RIDGE, int(alg_bytes))}
                     if hbm_bound else
                     {"bound": "mfma", "achieved": achieved, "peak": PEAK_TFLOPS[dtype], "unit": "TFLOP/s",
                      "frac": achieved / PEAK_TFLOPS[dtype], "traffic": hbm_traffic(args.config, mode),
                      "frac_steady": f_step / (steady_dev_ms / args.steps * 1e-3) / 1e12 / PEAK_TFLOPS[dtype],
                      "traffic_source": "stored: rocprofv3 PMC passes of the last profiled run of this workload (profiles/hbm_traffic.json, "
                                        "FETCH_SIZE x2 + WRITE_SIZE per launch), not measured in this run",
                      "effective_clock_ghz_profiled": stored_profile(args.config, mode).get("effective_clock_ghz"),
                      # stored telemetry of the last power-trace run of this workload (profiles/r5_ceiling_control.txt): socket power and
                      # power limit (hwmon), shader clock while the kernel runs back to back; and what this chip sustains on N(0,1)
                      # operands as a fraction of `peak` -- a RANGE with its sources (round 5): bare-MFMA loops of two shapes, and the
                      # vendor GEMM (hipBLASLt bf16 8192^3) as the control that does not depend on this build's own probe
                      **{k: stored_profile(args.config, mode).get(k) for k in ("power_w", "power_cap_w", "sclk_mhz",
                                                                                "mfma_only_random_frac_of_peak", "power_profile")},
                      "kernel_ms": kern_ms, **launch_stats(),
                      "note": "achieved = algorithmic FLOPs per step / HIP-event time per step on the launch stream (the `value` leg: "
                              "plain W + K protocol); frac_steady = the same for the conditioned leg; algorithmic bytes %d; the nominal "
                              "peak assumes 2.4 GHz, the chip sustains ~1.9 GHz under this kernel (effective_clock_ghz_profiled)" % int(alg_bytes)}),
    }

    if dist is not None and (n_gpus > 1 or os.environ.get("AULE_BENCH_FORCE_GATHER")):
        # the one exchange of the sharded path: every rank receives every rank's O over xGMI (RCCL), timed separately
        # from `value` (kernel-only scaling): (a) one blocking all-gather behind the whole shard (round 1), (b) the shard
        # in pieces, each piece's all-gather overlapping the next piece's kernels, (c) the same with direct peer sends
        # (all links of the fully connected node at once) -- aule/dist.py, DESIGN.md "Multi-GPU"
        from aule import dist as adist
        out = step().detach()
        gathered = torch.empty((n_gpus,) + tuple(out.shape), device=dev, dtype=out.dtype)

        def attn_nograd(a, b, c, causal=True, scale=None):
            return aule.flash_attention(a.detach(), b.detach(), c.detach(), causal=causal, scale=scale)

        def step_blocking():
            o = step().detach()
            dist.all_gather_into_tensor(gathered, o.contiguous())

        variants = {"blocking_all_gather": step_blocking}
        if mode == "fwd" and B >= 2:
            nch = min(4, B)
            variants["chunked_all_gather"] = lambda: adist.attention_and_gather(q, k, v, causal=causal, attn_fn=attn_nograd,
                                                                                  chunks=nch, transport="allgather")
            variants["chunked_p2p"] = lambda: adist.attention_and_gather(q, k, v, causal=causal, attn_fn=attn_nograd,
                                                                           chunks=nch, transport="p2p")
            # (d) no RCCL in the data path: every piece copied straight into the peers' mapped buffers (aule_peer_* of the
            # C-ABI, one hipMemcpyAsync per peer and piece on per-peer streams).  Its set-up is collective and agrees on
            # failure across the ranks, so a node where IPC mapping does not work reports the reason instead of hanging.
            variants["chunked_peer"] = lambda: adist.attention_and_gather(q, k, v, causal=causal, chunks=nch, transport="peer")
        gsteps = max(1, min(args.steps, 20))
        result["gather"] = {"bytes_per_rank": out.numel() * out.element_size(), "steps": gsteps}
        # The exchange variants are EXTRA to the contract's line and none of them has met N > 1 GPUs before the driver's run:
        # whatever happens in here, the line with `value` (measured above) gets printed.  A watchdog prints it and ends the
        # process if the section does not come back (a hung collective would otherwise take the whole scaling record with it);
        # an exception is recorded in the line.
        import threading
        section_done = threading.Event()
        limit_s = float(os.environ.get("AULE_BENCH_GATHER_TIMEOUT", "240"))

        def watchdog():
            if not section_done.wait(timeout=limit_s):
                result["gather"]["error"] = "the output-exchange section did not finish in %.0f s; line printed by the watchdog" % limit_s
                if rank == 0:
                    print(json.dumps(result), flush=True)
                os._exit(0)

        threading.Thread(target=watchdog, daemon=True).start()
        try:
            for name, fn in variants.items():
                try:
                    for _ in range(2):
                        fn()
                except aule.AuleError as e:     # (raised on every rank alike: aule.dist.PeerExchange)
                    result["gather"][name] = {"error": str(e)[:300]}
                    continue
                gwall, _ = timed(fn, gsteps)
                result["gather"][name] = {"ms_per_step": gwall * 1e3 / gsteps, "value": f_step * n_gpus * gsteps / gwall / 1e12,
                                          "exposed_ms": gwall * 1e3 / gsteps - ms_per_step}
            adist.release_peer_buffers()
        except Exception as e:   # noqa: BLE001 -- keep the contract's line
            result["gather"]["error"] = (type(e).__name__ + ": " + str(e))[:400]
        section_done.set()
        del gathered

    if rank == 0 and n_gpus == 1 and not args.no_extra and args.config == "c2":
        # the other half of the metric: fwd+bwd on config #3 (GQA 32q/8kv S=2048, B=4 as SURVEY 8d assumes), and config #5
        # (MQA 32q/1kv S=16384 D=64 fp16 non-causal forward), outside the timed region, the chip already conditioned
        B3, H3, K3, S3, _, D3, _, _, _ = CONFIGS["c3"]
        g3 = torch.Generator(device=dev).manual_seed(99)
        q3 = torch.randn(B3, H3, S3, D3, device=dev, dtype=torch.bfloat16, generator=g3).requires_grad_(True)
        k3 = torch.randn(B3, K3, S3, D3, device=dev, dtype=torch.bfloat16, generator=g3).requires_grad_(True)
        v3 = torch.randn(B3, K3, S3, D3, device=dev, dtype=torch.bfloat16, generator=g3).requires_grad_(True)
        d3 = torch.randn(B3, H3, S3, D3, device=dev, dtype=torch.bfloat16, generator=g3)

        def step3():
            q3.grad = k3.grad = v3.grad = None
            aule.flash_attention(q3, k3, v3, causal=True).backward(d3)

        def step3f():
            aule.flash_attention(q3, k3, v3, causal=True)

        condition(step3, args.condition_ms)   # its own steady state: right behind the C2 legs the chip is still power-limited by them
        _, ms3 = timed(step3, 30)
        l3 = sorted(last_launches)
        _, ms3f = timed(step3f, 30)
        f3f = fwd_flops(B3, H3, S3, S3, D3, True)
        t3 = 3.5 * f3f / (ms3 / 30 * 1e-3) / 1e12
        bwd_ms = (ms3 - ms3f) / 30
        result["extra"] = {"c3_fwd_bwd_tflops": t3, "c3_fwd_bwd_frac_of_peak": t3 / PEAK_TFLOPS["bf16"],
                           "c3_ms_per_step": ms3 / 30, "c3_ms_per_step_median": l3[len(l3) // 2],
                           "c3_fwd_ms": ms3f / 30, "c3_fwd_tflops": f3f / (ms3f / 30 * 1e-3) / 1e12,
                           "c3_bwd_ms": bwd_ms, "c3_bwd_tflops": 2.5 * f3f / (bwd_ms * 1e-3) / 1e12,
                           "c3_bwd_frac": 2.5 * f3f / (bwd_ms * 1e-3) / 1e12 / PEAK_TFLOPS["bf16"],
                           "c3_workload": "GQA 32q/8kv B=4 S=2048 D=128 bf16 causal fwd+bwd (autograd); bwd = step - fwd-only step "
                                          "(dQ / dK,dV kernel split: profiles/r3_fwdbwd_c3_*)"}
        del q3, k3, v3, d3
        # the metric's own fwd+bwd shape: C2 (B4 H32 S4096 D128 bf16 causal) through autograd, on this run's q, k, v
        # (the reference harness sweeps these shapes fwd+bwd: tests/benchmark_mi300x.py:207-233)
        d2 = torch.randn(B, Hq, Sq, D, device=dev, dtype=tdt, generator=g3)

        def step2():
            q.grad = k.grad = v.grad = None
            aule.flash_attention(q, k, v, causal=causal).backward(d2)

        condition(step2, args.condition_ms)
        _, ms2 = timed(step2, 20)
        l2 = sorted(last_launches)
        condition(step, args.condition_ms / 2)
        _, ms2f = timed(step, 20)
        t2 = 3.5 * f_fwd / (ms2 / 20 * 1e-3) / 1e12
        bwd2_ms = (ms2 - ms2f) / 20
        result["extra"].update({"c2_fwd_bwd_tflops": t2, "c2_fwd_bwd_frac_of_peak": t2 / PEAK_TFLOPS[dtype],
                                "c2_fwd_bwd_ms_per_step": ms2 / 20, "c2_fwd_bwd_ms_per_step_median": l2[len(l2) // 2],
                                "c2_fwd_ms": ms2f / 20, "c2_bwd_ms": bwd2_ms,
                                "c2_bwd_tflops": 2.5 * f_fwd / (bwd2_ms * 1e-3) / 1e12,
                                "c2_bwd_frac": 2.5 * f_fwd / (bwd2_ms * 1e-3) / 1e12 / PEAK_TFLOPS[dtype],
                                "c2_fwd_bwd_workload": "MHA B=4 H=32 S=4096 D=128 bf16 causal fwd+bwd (autograd): the metric's own shape; bwd = "
                                                       "step - fwd-only step (per-kernel split: profiles/r4_fwdbwd_c2_*)"})
        q.grad = k.grad = v.grad = None
        del d2
        B5, H5, K5, S5, _, D5, _, _, _ = CONFIGS["c5"]
        q5 = torch.randn(B5, H5, S5, D5, device=dev, dtype=torch.float16, generator=g3).requires_grad_(True)
        k5 = torch.randn(B5, K5, S5, D5, device=dev, dtype=torch.float16, generator=g3)
        v5 = torch.randn(B5, K5, S5, D5, device=dev, dtype=torch.float16, generator=g3)

        def step5():
            aule.flash_attention(q5, k5, v5, causal=False)

        condition(step5, args.condition_ms)
        _, ms5 = timed(step5, 20)
        t5 = fwd_flops(B5, H5, S5, S5, D5, False) / (ms5 / 20 * 1e-3) / 1e12
        result["extra"].update({"c5_fwd_tflops": t5, "c5_fwd_frac_of_peak": t5 / PEAK_TFLOPS["fp16"], "c5_ms_per_step": ms5 / 20,
                                "c5_workload": "MQA 32q/1kv B=1 S=16384 D=64 fp16 non-causal fwd (LSE stored)"})

        del q5, k5, v5
        # single-sequence prefill, the small-grid corner (128 paired items on 256 CUs: route 7, pairs of Q blocks cut into
        # key ranges + merge), and RoPE + attention the inference way (K by the pass, Q rotated inside the kernel, 3.6)
        q6, k6, v6 = (torch.randn(1, 8, 8192, 128, device=dev, dtype=torch.bfloat16, generator=g3) for _ in range(3))

        def step6():
            with torch.no_grad():
                aule.flash_attention(q6, k6, v6, causal=True)

        condition(step6, args.condition_ms)
        _, ms6 = timed(step6, 20)
        result["extra"].update({"b1h8_s8192_fwd_tflops": fwd_flops(1, 8, 8192, 8192, 128, True) / (ms6 / 20 * 1e-3) / 1e12,
                                "b1h8_s8192_ms_per_step": ms6 / 20,
                                "b1h8_s8192_workload": "MHA 8 heads B=1 S=8192 D=128 bf16 causal fwd (small grid, route 7: the one-wave-per-SIMD kernel over 256 key-range pieces + merge kernel)"})
        del q6, k6, v6
        B7, H7, S7, D7 = 4, 32, 2048, 128
        q7, k7, v7 = (torch.randn(B7, H7, S7, D7, device=dev, dtype=torch.bfloat16, generator=g3) for _ in range(3))
        cos7, sin7 = aule.precompute_rope_frequencies(S7, D7, device=dev)

        def step7():
            with torch.no_grad():
                aule.flash_attention_rope(q7, k7, v7, cos7, sin7, causal=True)

        condition(step7, args.condition_ms)
        _, ms7 = timed(step7, 20)
        result["extra"].update({"rope_attn_c2_tflops": fwd_flops(B7, H7, S7, S7, D7, True) / (ms7 / 20 * 1e-3) / 1e12,
                                "rope_attn_c2_ms_per_step": ms7 / 20,
                                "rope_attn_c2_workload": "RoPE + attention, B=4 H=32 S=2048 D=128 bf16 causal, inference: rope(K) pass + the "
                                                         "one-wave-per-SIMD forward with Q rotated inside it (attention FLOPs only)"})
        del q7, k7, v7
        # D = 64 training (the one-wave-per-SIMD backward pair's D = 64 instances, round 4) and the fp32 kernels (what the legacy
        # C-ABI and NumPy / fp32 torch input run; priced against the 157.3 TF f32-MFMA roof, not the bf16 peak)
        q8 = torch.randn(8, 32, 2048, 64, device=dev, dtype=torch.bfloat16, generator=g3).requires_grad_(True)
        k8 = torch.randn(8, 32, 2048, 64, device=dev, dtype=torch.bfloat16, generator=g3).requires_grad_(True)
        v8 = torch.randn(8, 32, 2048, 64, device=dev, dtype=torch.bfloat16, generator=g3).requires_grad_(True)
        d8 = torch.randn(8, 32, 2048, 64, device=dev, dtype=torch.bfloat16, generator=g3)

        def step8():
            q8.grad = k8.grad = v8.grad = None
            aule.flash_attention(q8, k8, v8, causal=True).backward(d8)

        def step8f():
            aule.flash_attention(q8, k8, v8, causal=True)

        condition(step8, args.condition_ms)
        _, ms8 = timed(step8, 20)
        _, ms8f = timed(step8f, 20)
        f8 = fwd_flops(8, 32, 2048, 2048, 64, True)
        b8 = (ms8 - ms8f) / 20
        result["extra"].update({"d64_fwd_bwd_tflops": 3.5 * f8 / (ms8 / 20 * 1e-3) / 1e12, "d64_bwd_ms": b8,
                                "d64_bwd_tflops": 2.5 * f8 / (b8 * 1e-3) / 1e12, "d64_bwd_frac": 2.5 * f8 / (b8 * 1e-3) / 1e12 / PEAK_TFLOPS["bf16"],
                                "d64_workload": "MHA B=8 H=32 S=2048 D=64 bf16 causal fwd+bwd (autograd); bwd = step - fwd-only step"})
        del q8, k8, v8, d8
        F32_ROOF = 157.3
        q9 = torch.randn(4, 32, 2048, 64, device=dev, dtype=torch.float32, generator=g3).requires_grad_(True)
        k9 = torch.randn(4, 32, 2048, 64, device=dev, dtype=torch.float32, generator=g3).requires_grad_(True)
        v9 = torch.randn(4, 32, 2048, 64, device=dev, dtype=torch.float32, generator=g3).requires_grad_(True)
        d9 = torch.randn(4, 32, 2048, 64, device=dev, dtype=torch.float32, generator=g3)

        def step9():
            q9.grad = k9.grad = v9.grad = None
            aule.flash_attention(q9, k9, v9, causal=True).backward(d9)

        def step9f():
            aule.flash_attention(q9, k9, v9, causal=True)

        condition(step9, args.condition_ms)
        _, ms9 = timed(step9, 10)
        _, ms9f = timed(step9f, 10)
        f9 = fwd_flops(4, 32, 2048, 2048, 64, True)
        b9 = (ms9 - ms9f) / 10
        result["extra"].update({"f32_fwd_tflops": f9 / (ms9f / 10 * 1e-3) / 1e12, "f32_fwd_frac_of_f32_roof": f9 / (ms9f / 10 * 1e-3) / 1e12 / F32_ROOF,
                                "f32_bwd_tflops": 2.5 * f9 / (b9 * 1e-3) / 1e12, "f32_bwd_frac_of_f32_roof": 2.5 * f9 / (b9 * 1e-3) / 1e12 / F32_ROOF,
                                "f32_roof_tflops": F32_ROOF,
                                "f32_workload": "MHA B=4 H=32 S=2048 D=64 fp32 causal (v_mfma_f32_32x32x2_f32 kernels: the legacy C-ABI's dtype), "
                                                "fwd-only step and fwd+bwd step (autograd); bwd = difference"})
        del q9, k9, v9, d9
        # The reference's OWN harness (python/aule/triton_flash_amd.py:775-813: fp16 causal forward, B 1 / 8, H 32, S 2048 / 8192, D 128, warm-up 10,
        # timed 50; tests/benchmark_mi300x.py:207-233 adds B 1 H 32 S 4096) and its comparator: the only number the reference publishes for
        # this path is relative to torch SDPA (python/README.md:20-23, "+6.1 .. +9.6 %").  Same process, same tensors, same protocol for both;
        # SDPA is a COMPARATOR here and nowhere else (never the product path).  FLOPs: the causal-discounted convention of `value`.
        import torch.nn.functional as F
        harness = []
        for (Bh, Hh, Sh) in ((1, 32, 2048), (8, 32, 2048), (1, 32, 8192), (1, 32, 4096)):
            qh, kh, vh = (torch.randn(Bh, Hh, Sh, 128, device=dev, dtype=torch.float16, generator=g3) for _ in range(3))

            def step_a():
                with torch.no_grad():
                    return aule.flash_attention(qh, kh, vh, causal=True)

            def step_s():
                with torch.no_grad():
                    return F.scaled_dot_product_attention(qh, kh, vh, is_causal=True)

            fl = fwd_flops(Bh, Hh, Sh, Sh, 128, True)
            row = {"shape": "B%d H%d S%d D128 fp16 causal fwd" % (Bh, Hh, Sh)}
            condition(step_a, args.condition_ms / 2)
            for _ in range(10):
                step_a()
            _, msa = timed(step_a, 50)
            row.update({"aule_ms": msa / 50, "aule_tflops": fl / (msa / 50 * 1e-3) / 1e12, "aule_tokens_per_s": Bh * Sh / (msa / 50 * 1e-3),
                        "aule_tflops_ref_convention": 4.0 * Bh * Hh * Sh * Sh * 128 / (msa / 50 * 1e-3) / 1e12})
            try:
                condition(step_s, args.condition_ms / 2)
                for _ in range(10):
                    step_s()
                _, mss = timed(step_s, 50)
                err = (step_a().float() - step_s().float()).abs().max().item()
                row.update({"sdpa_ms": mss / 50, "sdpa_tflops": fl / (mss / 50 * 1e-3) / 1e12, "speedup_vs_sdpa": mss / msa,
                            "max_abs_diff_vs_sdpa": err})
            except Exception as e:   # noqa: BLE001 -- the comparator must not take the line with it
                row["sdpa_error"] = (type(e).__name__ + ": " + str(e))[:200]
            harness.append(row)
            del qh, kh, vh
        result["extra"]["ref_harness"] = {
            "rows": harness,
            "protocol": "fp16 causal forward, no LSE (inference call), 10 warm-ups + 50 timed launches between HIP events after conditioning; "
                        "torch.nn.functional.scaled_dot_product_attention(is_causal=True) on the same tensors in the same process as the comparator "
                        "(torch %s; backend chosen by torch)" % torch.__version__,
            "reference_claim": "python/README.md:20-23: the reference's Triton kernel vs PyTorch SDPA on MI300X, +6.1 .. +9.6 %"}

        # Power: the same C2 forward on ALL-ZERO inputs (the same launch, the same instruction stream, the same MFMA count; no
        # data-dependent switching in the matrix pipes, the register files and LDS).  The distance between this figure and
        # `steady_state` is what the chip's power cap costs on N(0,1) data: the clock it sustains, not the kernel's schedule
        # (the backward kernels show the same ratio: tools/cbench.cpp with CB_AMP=0).
        qz, kz, vz = (torch.zeros(B, h, Sq, D, device=dev, dtype=tdt) for h in (Hq, Hkv, Hkv))

        def stepz():
            with torch.no_grad():
                aule.flash_attention(qz, kz, vz, causal=causal)

        condition(stepz, args.condition_ms)
        _, msz = timed(stepz, 20)
        tz = fwd_flops(B, Hq, Sq, Sk, D, causal) / (msz / 20 * 1e-3) / 1e12
        result["extra"].update({"c2_zero_inputs_fwd_tflops": tz, "c2_zero_inputs_frac_of_peak": tz / PEAK_TFLOPS[dtype],
                                "c2_zero_inputs_ms_per_step": msz / 20,
                                "c2_zero_inputs_note": "the C2 forward on all-zero q, k, v (LSE not stored): the kernel's schedule without the "
                                                       "data-dependent power of N(0,1) inputs -- NOT a throughput claim"})
        if isinstance(result.get("roofline"), dict):
            # beside frac / frac_steady: what the same kernel reaches of the peak when the power cap is out of the way
            result["roofline"]["frac_zero_inputs"] = tz / PEAK_TFLOPS[dtype]
        del qz, kz, vz

    if rank == 0:
        # the reference harness counts 4*B*H*S^2*D with NO causal discount (tests/benchmark_attention.zig:68-75):
        # for a causal shape that includes the masked half, so it is about twice `value`.  Printed for comparison with the
        # reference's own reports only, labelled, and never used for `value` or the roofline.
        f_ref = 4.0 * B * Hq * Sq * Sk * D * (3.5 if mode == "fwdbwd" else 1.0)
        result["ref_harness_convention"] = {
            "tflops": f_ref * n_gpus * args.steps / wall / 1e12,
            "formula": "4*B*Hq*Sq*Sk*D (x3.5 fwd+bwd), no causal discount -- counts masked work; NOT achieved throughput",
            "ratio_to_value": f_ref / f_step}

    if rank == 0 and n_gpus == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline()

    if rank == 0:
        print(json.dumps(result))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
