"""Multi-GPU execution of the hot path: batch x kv-head sharding, no collective inside the attention computation, and ONE
exchange -- the gather of the output -- chunked and overlapped with the kernels that are still running.

The reference is single-device (no NCCL/RCCL call site anywhere); this is the MI355X-native scale-out
named by BASELINE.json's north_star and specified in SURVEY.md 8(e): one process per GPU, torch.distributed with backend
"nccl" (= RCCL over xGMI).

Partitioning.  Every (batch, kv-head) unit is independent, so units are split across ranks: along the batch when
B >= world, otherwise along the flattened (batch, kv-head) axis; a KV head and the query heads of its group always stay on
one rank (no K/V duplication, no K/V traffic at all).  Shard sizes differ by at most one unit.

The gather.  A rank's output is `units/world * g * Sq * D * elt` bytes (config 4: 512 MiB of 4 GiB).  On the 8-GPU node
every GPU pair has its own xGMI link (7 links x ~153 GB/s per GPU), so
  * a ring all-gather moves (n-1)/n of the tensor over ONE link per hop: config 4, 3.76 GB / 153 GB/s = 24.5 ms;
  * a direct exchange (every rank sends its shard to all 7 peers at once) uses all 7 links: 0.5 GiB / 153 GB/s = 3.5 ms
against 3.8 ms of kernel time per rank.  So (1) the local shard is computed in `chunks` pieces along its leading axis
(contiguous views: no copies, no change to the kernels -- each piece is an ordinary batch of independent heads), and the
gather of piece i is launched asynchronously (RCCL runs it on its own stream, ordered behind the kernel that produced
it) while piece i+1 computes; (2) `transport="p2p"` posts the direct sends / receives of a piece to all peers as one
batch (torch.distributed.batch_isend_irecv -> one RCCL group: all links busy, and shards of different sizes need no
padding), `transport="allgather"` posts one all-gather per piece into per-rank views of the final tensor.  Expected
end-to-end cost of the gather at config 4: ~0.4 ms exposed (the last piece) instead of 3.5 ms (p2p) / 24.5 ms (ring).
Every rank ends with the full [B, Hq, Sq, D] tensor; pass gather=False to keep the shard (what a data-parallel model does:
it never needs the other ranks' attention outputs).  The backward needs no collective at all: dQ, dK, dV are per unit.
"""
from typing import Callable, List, Optional, Tuple


def partition(n_units: int, world: int) -> List[Tuple[int, int]]:
    """Contiguous [start, end) ranges of `n_units` for each rank, sizes differing by <= 1."""
    base, rem = divmod(n_units, world)
    out, s = [], 0
    for r in range(world):
        e = s + base + (1 if r < rem else 0)
        out.append((s, e))
        s = e
    return out


def shard_plan(batch: int, heads_kv: int, world: int):
    """How to split [B, H, S, D] tensors: along batch when B >= world, otherwise along the
    flattened (batch, kv-head) axis.  Returns ("batch"|"unit", ranges)."""
    if batch >= world:
        return "batch", partition(batch, world)
    return "unit", partition(batch * heads_kv, world)


def local_shard(q, k, v, rank: int, world: int):
    """Views of this rank's share of q, k, v (torch tensors, [B,H,S,D])."""
    B, Hq = q.shape[0], q.shape[1]
    Hkv = k.shape[1]
    g = Hq // Hkv
    mode, ranges = shard_plan(B, Hkv, world)
    s, e = ranges[rank]
    if mode == "batch":
        return mode, q[s:e], k[s:e], v[s:e]
    # flatten (b, hkv) -> units; q heads of unit u are [u*g, (u+1)*g) in the flattened (b, hq) axis
    Sq, Sk, D = q.shape[2], k.shape[2], q.shape[3]
    qf = q.reshape(B * Hkv, g, Sq, D)[s:e].reshape(1, (e - s) * g, Sq, D)
    kf = k.reshape(B * Hkv, 1, Sk, D)[s:e].reshape(1, e - s, Sk, D)
    vf = v.reshape(B * Hkv, 1, Sk, D)[s:e].reshape(1, e - s, Sk, D)
    return mode, qf, kf, vf


def chunk_ranges(n: int, chunks: int) -> List[Tuple[int, int]]:
    """Split range(n) into at most `chunks` contiguous non-empty pieces of near-equal size."""
    chunks = max(1, min(int(chunks), n))
    return [r for r in partition(n, chunks) if r[1] > r[0]] if n > 0 else []


def gather_bytes(batch: int, heads_q: int, heads_kv: int, seq_q: int, head_dim: int, elt: int, world: int):
    """Bytes each rank contributes to / receives from the output exchange: (send_per_peer, recv_total)."""
    _, ranges = shard_plan(batch, heads_kv, world)
    mode = "batch" if batch >= world else "unit"
    per_unit = (heads_q if mode == "batch" else heads_q // heads_kv) * seq_q * head_dim * elt
    sizes = [(e - s) * per_unit for s, e in ranges]
    return sizes, sum(sizes)


def flash_attention_sharded(q, k, v, causal: bool = True, scale: Optional[float] = None, group=None,
                            gather: bool = True, attn_fn: Optional[Callable] = None, chunks: int = 4,
                            transport: str = "auto"):
    """Every rank holds the same full q, k, v (or at least its own shard's rows); each computes its share with `attn_fn`
    (default aule.flash_attention) in `chunks` pieces and, if `gather`, every rank receives the full output: the exchange of
    piece i overlaps the computation of piece i+1 (module docstring).  transport: "allgather", "p2p" or "auto" (p2p when
    the shards differ in size, all-gather otherwise).  Returns the full [B,Hq,Sq,D] output (gather=True) or this rank's
    shard.  Inference path (no autograd through the collective).  chunks=1, transport="allgather" is the single blocking
    collective of round 1."""
    import torch
    import torch.distributed as dist
    if attn_fn is None:
        from . import flash_attention as attn_fn
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    B, Hq, Sq, D = q.shape
    Hkv = k.shape[1]
    g = Hq // Hkv
    mode, qs, ks, vs = local_shard(q, k, v, rank, world)
    _, ranges = shard_plan(B, Hkv, world)
    # rows of the flattened [B*Hq, Sq, D] output each rank owns; a "piece" is a range of the shard's leading axis
    lead = 0 if mode == "batch" else 1                      # batch items, or heads of the single flattened batch item
    per_lead = Hq if mode == "batch" else 1                 # flattened output rows per leading index
    units_to_lead = 1 if mode == "batch" else g
    n_lead = [(e - s) * units_to_lead for s, e in ranges]   # leading extent per rank
    row0 = [s * units_to_lead * per_lead for s, _ in ranges]

    def run(sl):
        qq = qs[sl] if lead == 0 else qs[:, sl]
        if lead == 0:
            kk, vv = ks[sl], vs[sl]
        else:   # heads of one flattened batch item: the KV heads of the same units
            kk, vv = ks[:, sl.start // g:(sl.stop + g - 1) // g], vs[:, sl.start // g:(sl.stop + g - 1) // g]
        return attn_fn(qq.contiguous(), kk.contiguous(), vv.contiguous(), causal=causal, scale=scale)

    if not gather:
        if n_lead[rank] == 0:
            return qs.new_empty(qs.shape)
        return run(slice(0, n_lead[rank]))
    full = _compute_and_exchange(run, n_lead, row0, per_lead, g if lead == 1 else 1, rank, world, (B * Hq, Sq, D),
                                 q.dtype, q.device, chunks, transport, group)
    return full.reshape(B, Hq, Sq, D)


def attention_and_gather(q, k, v, causal: bool = True, scale: Optional[float] = None, group=None,
                         attn_fn: Optional[Callable] = None, chunks: int = 4, transport: str = "allgather"):
    """The same exchange for ranks that hold ONLY their own shard (equal shards along the batch axis: what a data-parallel
    caller and bench.py have): q [Bl,Hq,Sq,D], k / v [Bl,Hkv,Sk,D] local; returns [world*Bl,Hq,Sq,D] on every rank,
    rank r's rows at [r*Bl, (r+1)*Bl)."""
    import torch.distributed as dist
    if attn_fn is None:
        from . import flash_attention as attn_fn
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    Bl, Hq, Sq, D = q.shape

    def run(sl):
        return attn_fn(q[sl].contiguous(), k[sl].contiguous(), v[sl].contiguous(), causal=causal, scale=scale)

    full = _compute_and_exchange(run, [Bl] * world, [r * Bl * Hq for r in range(world)], Hq, 1, rank, world,
                                 (world * Bl * Hq, Sq, D), q.dtype, q.device, chunks, transport, group)
    return full.reshape(world * Bl, Hq, Sq, D)


def _compute_and_exchange(run, n_lead, row0, per_lead, step, rank, world, shape, dtype, device, chunks, transport, group):
    """run(slice over the shard's leading axis) -> that piece's output; n_lead[r] = leading extent of rank r's shard,
    row0[r] = its first row in the flattened [rows, Sq, D] result, per_lead = rows per leading index, step = granularity of
    a piece (a query group stays whole).  Computes this rank's pieces and exchanges them, piece i's exchange overlapping
    piece i+1's kernels."""
    import torch
    import torch.distributed as dist
    Sq, D = shape[1], shape[2]
    if transport == "auto":
        transport = "allgather" if len(set(n_lead)) == 1 else "p2p"
    if transport not in ("allgather", "p2p"):
        raise ValueError(f"transport must be 'auto', 'allgather' or 'p2p', got {transport!r}")
    if transport == "allgather" and len(set(n_lead)) != 1:
        raise ValueError("transport='allgather' needs equal shards; use 'p2p' (or 'auto') for a ragged split")

    full = torch.empty(shape, dtype=dtype, device=device)
    works = []
    if transport == "allgather":
        pieces = [(a * step, b * step) for a, b in chunk_ranges(n_lead[rank] // step, chunks)]
        for a, b in pieces:
            piece = run(slice(a, b)).reshape(-1, Sq, D)
            views = [full[row0[r] + a * per_lead: row0[r] + b * per_lead] for r in range(world)]
            works.append(dist.all_gather(views, piece, group=group, async_op=True))
    else:
        # every rank cuts ITS shard into the same number of pieces (possibly of different sizes); piece i of every rank
        # is exchanged in one batch of sends / receives
        npiece = max(1, min(int(chunks), max(1, min(n // step for n in n_lead if n > 0) if any(n_lead) else 1)))
        cuts = [[(a * step, b * step) for a, b in (partition(n // step, npiece) if n > 0 else [(0, 0)] * npiece)]
                for n in n_lead]
        for i in range(npiece):
            a, b = cuts[rank][i]
            piece = run(slice(a, b)).reshape(-1, Sq, D) if b > a else None
            if piece is not None:
                full[row0[rank] + a * per_lead: row0[rank] + b * per_lead].copy_(piece)
            ops = []
            for r in range(world):
                if r == rank:
                    continue
                ra, rb = cuts[r][i]
                if rb > ra:
                    ops.append(dist.P2POp(dist.irecv, full[row0[r] + ra * per_lead: row0[r] + rb * per_lead], r, group))
                if piece is not None:
                    ops.append(dist.P2POp(dist.isend, piece, r, group))
            if ops:
                works.extend(dist.batch_isend_irecv(ops))
    for w in works:
        w.wait()
    return full
