"""bench.py's launch contract on CPU: `python bench.py --gpus N` without a launcher spawns
its own ranks the way the driver does (torch.distributed.run, 127.0.0.1), rank 0 prints ONE JSON line; without enough devices
it fails with that message and not with launch instructions.  --dry-run = the scaffolding only (gloo, no device, no kernels)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*args, env=None):
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(args), cwd=ROOT, env=e, capture_output=True,
                          text=True, timeout=300)


def _json_lines(out):
    return [json.loads(l) for l in out.splitlines() if l.startswith("{")]


def test_plain_launch_with_two_ranks_spawns_itself():
    r = _run("--gpus", "2", "--dry-run", "--steps", "4", "--warmup", "1")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1, r.stdout
    assert lines[0]["n_gpus"] == 2 and lines[0]["world_size"] == 2 and lines[0]["dry_run"] is True
    assert lines[0]["steps"] == 4 and lines[0]["warmup"] == 1 and lines[0]["value"] is None
    # N > 1 times BASELINE.json configs[3] (B=64 over 8 GPUs: 8 per GPU, S=8192), not the single-GPU headline's batch
    assert lines[0]["config"]["workload"].startswith("c4: B=8/GPU Hq=32 Hkv=32 Sq=8192 Sk=8192 D=128 bf16 causal fwd"), lines[0]["config"]


def test_dry_run_one_rank_needs_no_launcher():
    r = _run("--gpus", "1", "--dry-run", "--steps", "2", "--warmup", "0")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1 and lines[0]["world_size"] == 1
    assert lines[0]["config"]["workload"].startswith("c2: B=4/GPU Hq=32 Hkv=32 Sq=4096 Sk=4096 D=128 bf16 causal fwd"), lines[0]["config"]


def test_launcher_with_wrong_rank_count_is_refused():
    # the driver's command shape with 2 ranks but --gpus 3: every rank refuses, nothing hangs
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", "29531", os.path.join(ROOT, "bench.py"), "--gpus", "3", "--dry-run"],
                       cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "--gpus 3 but the launcher made 2 rank(s)" in r.stderr + r.stdout


def test_without_devices_the_message_names_the_device_count():
    import torch
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        import pytest
        pytest.skip("this box has two devices")
    r = _run("--gpus", "2", "--steps", "1", "--warmup", "0")
    assert r.returncode != 0
    assert "needs 2 devices" in r.stderr + r.stdout
    assert "torch.distributed.run" not in r.stderr + r.stdout


def test_plain_launch_with_eight_ranks_names_c4_and_its_shards():
    """The driver's scaling run at its widest: 8 ranks rendezvous on 127.0.0.1, time, reduce MAX over ranks, and
    rank 0 prints one line for BASELINE configs[3] -- B = 64 over 8 GPUs, 8 per GPU, 512 MiB of output per rank, 4 GiB gathered."""
    r = _run("--gpus", "8", "--dry-run", "--steps", "3", "--warmup", "1")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1, r.stdout
    rec = lines[0]
    assert rec["n_gpus"] == 8 and rec["world_size"] == 8 and rec["dry_run"] is True and rec["steps"] == 3
    assert rec["config"]["workload"].startswith("c4: B=8/GPU Hq=32 Hkv=32 Sq=8192 Sk=8192 D=128 bf16 causal fwd"), rec["config"]
    assert rec["gather_plan"] == {"bytes_per_rank": [512 << 20] * 8, "total_bytes": 4 << 30, "shard": "batch"}


def test_a_hung_exchange_section_still_prints_the_line():
    """The output-exchange variants have never met N > 1 GPUs: if one of them hangs (here: a barrier rank 1 never joins), the watchdog
    prints the contract's line with the reason and the process ends with status 0 -- the scaling record survives."""
    r = _run("--gpus", "2", "--dry-run", "--steps", "2", "--warmup", "0", env={"AULE_BENCH_TEST_HANG": "1", "AULE_BENCH_GATHER_TIMEOUT": "3"})
    assert r.returncode == 0, r.stderr[-2000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1, r.stdout
    assert lines[0]["world_size"] == 2 and "error" in lines[0]["gather"], lines[0]
