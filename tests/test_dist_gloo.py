"""World-size-2 (and 3) CPU tests (gloo) of the multi-GPU path: sharding + the chunked, overlapped output exchange -- as
per-piece all-gathers or as direct peer sends -- reproduce the unsharded result exactly, for every chunk count.  The per-shard compute function is
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

    def attn(a, b, c, causal=True, scale=None, window_size=-1):
        o, _ = oracle.fwd_f64(a.numpy(), b.numpy(), c.numpy(), causal, scale, window_size)
        return torch.from_numpy(o)

    ref = attn(tq, tk, tv, causal=causal)
    ok = True
    # (round 6) the sliding window passes through the sharded entry point: every shard is a set of whole heads
    if causal:
        W = max(2, Sq // 3)
        wref = attn(tq, tk, tv, causal=causal, window_size=W)
        ok = ok and not bool(torch.equal(wref, ref))
        ok = ok and bool(torch.equal(adist.flash_attention_sharded(tq, tk, tv, causal=causal, attn_fn=attn, chunks=2, window_size=W), wref))
    equal_shards = len({e - s for s, e in adist.shard_plan(B, Hkv, world)[1]}) == 1
    for transport in ("auto", "p2p") + (("allgather",) if equal_shards else ()):
        for chunks in (1, 2, 3, 8):
            full = adist.flash_attention_sharded(tq, tk, tv, causal=causal, attn_fn=attn, chunks=chunks, transport=transport)
            ok = ok and bool(torch.equal(full, ref))    # chunked == unchunked == unsharded, bit for bit
    if not equal_shards:
        try:
            adist.flash_attention_sharded(tq, tk, tv, causal=causal, attn_fn=attn, transport="allgather")
            ok = False                                  # a ragged split must be refused by the all-gather transport
        except ValueError:
            pass
    try:
        adist.flash_attention_sharded(tq, tk, tv, causal=causal, attn_fn=attn, transport="peer")
        ok = False                                      # the peer transport exchanges device buffers: host tensors are refused
    except ValueError:
        pass
    shard = adist.flash_attention_sharded(tq, tk, tv, causal=causal, gather=False, attn_fn=attn)
    q.put((rank, ok, tuple(shard.shape)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("case,world", [
    ((4, 4, 2, 24, 24, 16, True), 2),     # batch split 2+2
    ((8, 4, 2, 16, 16, 16, True), 2),     # batch split 4+4: up to four pieces per rank
    ((3, 4, 2, 16, 20, 16, False), 2),    # ragged batch split 2+1: direct sends, no padding
    ((1, 6, 2, 16, 16, 16, True), 2),     # B < world: split (batch, kv-head) units, groups stay together
    ((1, 8, 1, 12, 12, 16, True), 2),     # single unit: rank 1 gets nothing
    ((2, 12, 6, 16, 16, 16, True), 3),    # three ranks, unit split 4+4+4 with groups of two query heads
    ((7, 2, 2, 16, 16, 16, False), 3),    # three ranks, ragged 3+2+2
])
def test_sharded_equals_unsharded(case, world):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, case, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
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
    for n in range(0, 12):
        for c in (1, 2, 5, 20):
            r = adist.chunk_ranges(n, c)
            assert (not r and n == 0) or (r[0][0] == 0 and r[-1][1] == n and all(b > a for a, b in r) and len(r) <= max(1, c))
    # config 4: B=64 H=32 S=8192 D=128 bf16 over 8 ranks -> 512 MiB per rank, 4 GiB in total
    sizes, total = adist.gather_bytes(64, 32, 32, 8192, 128, 2, 8)
    assert sizes == [512 << 20] * 8 and total == 4 << 30


def _worker_local(rank, world, port, q):
    """attention_and_gather: every rank holds ONLY its own equal shard (what bench.py --gpus N does) and ends with the
    concatenation of all shards' outputs, for both transports and several piece counts."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "aule-attention_amd"))
    import torch
    import torch.distributed as dist
    import oracle
    from aule import dist as adist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    Bl, Hq, Hkv, S, D = 3, 4, 2, 20, 16

    def shard(r):
        rng = np.random.RandomState(100 + r)
        return tuple(torch.from_numpy(rng.randn(*s).astype(np.float32)) for s in ((Bl, Hq, S, D), (Bl, Hkv, S, D), (Bl, Hkv, S, D)))

    def attn(a, b, c, causal=True, scale=None):
        o, _ = oracle.fwd_f64(a.numpy(), b.numpy(), c.numpy(), causal, scale)
        return torch.from_numpy(o)

    want = torch.cat([attn(*shard(r)) for r in range(world)], dim=0)
    mine = shard(rank)
    ok = True
    for transport in ("allgather", "p2p"):
        for chunks in (1, 2, 3, 7):
            full = adist.attention_and_gather(*mine, causal=True, attn_fn=attn, chunks=chunks, transport=transport)
            ok = ok and tuple(full.shape) == (world * Bl, Hq, S, D) and bool(torch.equal(full, want))
    q.put((rank, ok))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3, 8])
def test_local_shards_gathered(world):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_local, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok in res), res


@pytest.mark.parametrize("B,Hq,Hkv,world", [(8, 32, 32, 8), (4, 4, 2, 2), (3, 4, 2, 2), (1, 6, 2, 2), (1, 8, 1, 2), (2, 12, 6, 3),
                                            (7, 2, 2, 3), (64, 32, 32, 8), (5, 8, 4, 8)])
@pytest.mark.parametrize("chunks", [1, 3, 4])
def test_peer_copy_plans_tile_the_gathered_tensor(B, Hq, Hkv, world, chunks):
    """transport="peer": rank r copies piece (offset, bytes) of ITS buffer to the same offset of every peer's buffer.  The
    ranks' plans must tile the [B*Hq, Sq, D] tensor exactly once, pieces must lie inside the rank's own rows, and a query
    group must never be cut (the offset arithmetic of the exchange, without a device)."""
    sys.path.insert(0, os.path.join(ROOT, "aule-attention_amd"))
    from aule import dist as adist
    Sq, D, elt = 48, 64, 2
    row_bytes = Sq * D * elt
    mode, n_lead, row0, per_lead, step = adist.shard_layout(B, Hq, Hkv, world)
    covered = []
    for r in range(world):
        plan = adist.peer_copy_plan(n_lead, row0, per_lead, step, r, chunks, row_bytes)
        assert len(plan) <= max(1, chunks)
        lo, hi = row0[r] * row_bytes, (row0[r] + n_lead[r] * per_lead) * row_bytes
        for a, b, off, nb in plan:
            assert b > a and a % step == 0 and b % step == 0
            assert lo <= off and off + nb <= hi and nb == (b - a) * per_lead * row_bytes
            covered.append((off, off + nb))
    covered.sort()
    pos = 0
    for a, b in covered:
        assert a == pos, (covered, pos)
        pos = b
    assert pos == B * Hq * row_bytes


def test_config4_sharding_arithmetic_at_world_8():
    """BASELINE configs[3] as the driver's 8-GPU run shards it -- sizes only, no tensors: B = 64 over 8
    ranks is 8 batch elements = 512 MiB of output per rank; bench.py exchanges it in 4 pieces of 2 elements = 128 MiB each; every rank's
    copy plan (transport "peer") writes its 4 pieces at rank * 512 MiB + piece * 128 MiB of every peer's 4 GiB buffer; the 32 pieces
    tile the gathered tensor exactly; weak-scaling worlds 2 and 4 keep 8 per rank."""
    sys.path.insert(0, os.path.join(ROOT, "aule-attention_amd"))
    from aule import dist as adist
    B, Hq, Hkv, S, D, elt, world = 64, 32, 32, 8192, 128, 2, 8
    mode, bounds = adist.shard_plan(B, Hkv, world)
    assert mode == "batch" and bounds == [(8 * r, 8 * r + 8) for r in range(world)]
    sizes, total = adist.gather_bytes(B, Hq, Hkv, S, D, elt, world)
    assert sizes == [512 << 20] * world and total == 4 << 30
    assert adist.chunk_ranges(8, 4) == [(0, 2), (2, 4), (4, 6), (6, 8)]
    row_bytes = S * D * elt                                  # one (batch, head) row block of the output: 2 MiB
    lay_mode, n_lead, row0, per_lead, step = adist.shard_layout(B, Hq, Hkv, world)
    assert lay_mode == "batch" and n_lead == [8] * world and per_lead == Hq and step == 1
    pieces = []
    for r in range(world):
        assert row0[r] * row_bytes == r * (512 << 20)
        plan = adist.peer_copy_plan(n_lead, row0, per_lead, step, r, 4, row_bytes)
        assert [(a, b) for a, b, _, _ in plan] == [(0, 2), (2, 4), (4, 6), (6, 8)]
        for i, (_, _, off, nb) in enumerate(plan):
            assert nb == 128 << 20 and off == r * (512 << 20) + i * (128 << 20)
            pieces.append((off, nb))
    pieces.sort()
    assert [o for o, _ in pieces] == [i * (128 << 20) for i in range(32)] and sum(n for _, n in pieces) == 4 << 30
    for w in (2, 4):                                         # the scaling run's other points: 8 per rank again
        assert adist.shard_plan(8 * w, Hkv, w)[1] == [(8 * r, 8 * r + 8) for r in range(w)]
        assert adist.gather_bytes(8 * w, Hq, Hkv, S, D, elt, w) == ([512 << 20] * w, w * (512 << 20))
