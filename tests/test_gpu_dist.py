"""GPU test of the multi-GPU driver code on ONE GPU (the box has one): a 1-rank RCCL process group
runs aule.dist.flash_attention_sharded and bench.py's torch.distributed path end to end, so that
init / all_reduce / all_gather_into_tensor / barrier are exercised before the driver's 8-GPU run.
(The sharding arithmetic itself is covered with world_size 2 on gloo in tests/test_dist_gloo.py.)"""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _run(cmd, extra_env=None):
    env = dict(os.environ)
    env.update({"MASTER_ADDR": "127.0.0.1", "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    env.update(extra_env or {})
    return subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)


def test_bench_under_torchrun_one_rank():
    r = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
              "--master-addr", "127.0.0.1", "--master-port", "29517", "bench.py", "--gpus", "1", "--steps", "3",
              "--warmup", "1", "--batch", "2", "--condition-ms", "20", "--no-cpu-baseline", "--no-extra"],
             {"AULE_BENCH_FORCE_GATHER": "1"})
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    rec = json.loads(line)
    assert rec["n_gpus"] == 1 and rec["value"] > 0 and rec["unit"] == "TFLOP/s"
    # the output exchange, three ways (one blocking all-gather; pieces overlapped with the kernels, as all-gathers and as
    # direct peer sends): each timed end to end
    for name in ("blocking_all_gather", "chunked_all_gather", "chunked_p2p", "chunked_peer"):
        assert rec["gather"][name]["ms_per_step"] > 0, (name, rec["gather"][name])
    assert rec["steady_state"]["value"] > 0 and rec["steady_state"]["conditioning_steps"] > 0 and rec["world_size"] == 1
    assert rec["roofline"]["bound"] == "mfma" and 0 < rec["roofline"]["frac"] < 1


def test_sharded_api_one_rank_nccl():
    code = r"""
import os, sys
sys.path.insert(0, os.path.join(%r, "aule-attention_amd"))
import torch, torch.distributed as dist
import aule
from aule import dist as adist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29518")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
q = torch.randn(2, 8, 256, 128, device="cuda", dtype=torch.bfloat16)
k = torch.randn(2, 2, 256, 128, device="cuda", dtype=torch.bfloat16)
v = torch.randn(2, 2, 256, 128, device="cuda", dtype=torch.bfloat16)
ref = aule.flash_attention(q, k, v, causal=True)
for transport in ("auto", "allgather", "p2p", "peer"):
    for chunks in (1, 2, 4):
        full = adist.flash_attention_sharded(q, k, v, causal=True, chunks=chunks, transport=transport)
        assert torch.equal(full, ref), (transport, chunks)
full = adist.attention_and_gather(q, k, v, causal=True, chunks=2)
assert torch.equal(full, ref)
qw, kw, vw = (torch.randn(2, h, 1024, 128, device="cuda", dtype=torch.bfloat16) for h in (8, 2, 2))
wref = aule.flash_attention(qw, kw, vw, causal=True, window_size=256)      # (round 6: the window instances behind the sharded entry points)
assert not torch.equal(wref, aule.flash_attention(qw, kw, vw, causal=True))
for transport in ("allgather", "peer"):
    assert torch.equal(adist.flash_attention_sharded(qw, kw, vw, causal=True, chunks=2, transport=transport, window_size=256), wref), transport
assert torch.equal(adist.attention_and_gather(qw, kw, vw, causal=True, chunks=2, window_size=256), wref)
dist.destroy_process_group()
print("OK")
""" % ROOT
    r = _run([sys.executable, "-c", code])
    assert r.returncode == 0 and "OK" in r.stdout, r.stderr[-2000:]


@pytest.mark.parametrize("world", [2, 4])
def test_peer_exchange_two_processes_on_one_gpu(world):
    """transport="peer" end to end with TWO (round 6: and FOUR -- ranks that hold a single element, or nothing) ranks (the box has one GPU: both use cuda:0; gloo carries the handles and the
    final meeting, the payload goes through aule_peer_alloc / _open / _copy_async): every rank ends with the same gathered
    tensor as the unsharded call, for equal and ragged shards, twice in a row (the cached buffers alternate).  The ranks
    are tests/peer_exchange_worker.py."""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ)
    env.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "peer_exchange_worker.py"), str(r), str(world)], cwd=ROOT, env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(world)]
    outs = [p.communicate(timeout=600) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0 and "RANK_OK" in so, (so[-1500:], se[-2500:])


def test_bench_watchdog_prints_the_line_when_the_exchange_hangs():
    """bench.py's real (device) path: the output-exchange section never comes back (test hook) -> the watchdog prints the contract's
    line -- `value` measured before the section, the legs, the reason under gather.error -- and the process ends with status 0."""
    r = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
              "--master-addr", "127.0.0.1", "--master-port", "29519", "bench.py", "--gpus", "1", "--steps", "3",
              "--warmup", "1", "--batch", "1", "--condition-ms", "10", "--no-cpu-baseline", "--no-extra"],
             {"AULE_BENCH_FORCE_GATHER": "1", "AULE_BENCH_TEST_HANG": "1", "AULE_BENCH_GATHER_TIMEOUT": "4"})
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    rec = json.loads(lines[0])
    assert rec["value"] > 0 and "did not finish" in rec["gather"]["error"] and "value" in rec["legs"]
