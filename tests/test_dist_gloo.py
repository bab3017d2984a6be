"""World-size-2 CPU test (gloo) of the multi-GPU path: sharding + the single output
all-gather reproduce the unsharded result exactly.  The per-shard compute function is
injected (the oracle, as the checker) because there is no GPU here; on the GPU box the
same driver code runs with aule.flash_attention over RCCL (tests/test_gpu_dist.py)."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, case, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "aule-attention_amd"))
    import torch
    import torch.distributed as dist
    import oracle
    from aule import dist as adist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    B, Hq, Hkv, Sq, Sk, D, causal = case
    rng = np.random.RandomState(7)
    tq = torch.from_numpy(rng.randn(B, Hq, Sq, D).astype(np.float32))
    tk = torch.from_numpy(rng.randn(B, Hkv, Sk, D).astype(np.float32))
    tv = torch.from_numpy(rng.randn(B, Hkv, Sk, D).astype(np.float32))

    def attn(a, b, c, causal=True, scale=None):
        o, _ = oracle.fwd_f64(a.numpy(), b.numpy(), c.numpy(), causal, scale)
        return torch.from_numpy(o)

    full = adist.flash_attention_sharded(tq, tk, tv, causal=causal, attn_fn=attn)
    ref = attn(tq, tk, tv, causal=causal)
    ok = bool(torch.equal(full, ref))
    shard = adist.flash_attention_sharded(tq, tk, tv, causal=causal, gather=False, attn_fn=attn)
    q.put((rank, ok, tuple(shard.shape)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("case", [
    (4, 4, 2, 24, 24, 16, True),     # batch split 2+2
    (3, 4, 2, 16, 20, 16, False),    # ragged batch split 2+1 (padded gather)
    (1, 6, 2, 16, 16, 16, True),     # B < world: split (batch, kv-head) units, groups stay together
    (1, 8, 1, 12, 12, 16, True),     # single unit: rank 1 gets nothing
])
def test_sharded_equals_unsharded_world2(case):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, case, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res), res


def test_partition_properties():
    sys.path.insert(0, os.path.join(ROOT, "aule-attention_amd"))
    from aule import dist as adist
    for n in range(0, 20):
        for w in (1, 2, 3, 8):
            r = adist.partition(n, w)
            assert r[0][0] == 0 and r[-1][1] == n and all(a[1] == b[0] for a, b in zip(r, r[1:]))
            assert max(e - s for s, e in r) - min(e - s for s, e in r) <= 1
    assert adist.shard_plan(64, 32, 8)[0] == "batch" and adist.shard_plan(1, 8, 8)[0] == "unit"
