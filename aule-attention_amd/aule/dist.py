"""Multi-GPU execution of the hot path: batch x kv-head sharding + one output all-gather.

The reference is single-device (no NCCL/RCCL call site anywhere);
this is the MI355X-native scale-out named by BASELINE.json's north_star: every
(batch, kv-head) unit is independent, so units are split across ranks (one process per
GPU, torch.distributed with backend "nccl" = RCCL over xGMI) with NO collective inside the
attention computation; the only exchange is an optional all-gather of the output.
A KV head and the query heads of its group always stay on one rank (no K/V duplication).
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


def flash_attention_sharded(q, k, v, causal: bool = True, scale: Optional[float] = None, group=None,
                            gather: bool = True, attn_fn: Optional[Callable] = None):
    """Every rank holds the same full q, k, v (or at least its own shard's rows); each computes
    its share with `attn_fn` (default aule.flash_attention) and, if `gather`, all ranks receive the
    full output through ONE all-gather.  Returns the full [B,Hq,Sq,D] output (gather=True) or this
    rank's shard.  Inference path (no autograd through the collective)."""
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
    if qs.shape[0] * qs.shape[1] > 0:
        out_local = attn_fn(qs.contiguous(), ks.contiguous(), vs.contiguous(), causal=causal, scale=scale)
    else:
        out_local = qs.new_empty(qs.shape)
    if not gather:
        return out_local
    _, ranges = shard_plan(B, Hkv, world)
    rows = [(e - s) * (Hq if mode == "batch" else g) for s, e in ranges]
    flat_local = out_local.reshape(-1, Sq, D).contiguous()
    if len(set(rows)) == 1:
        full = torch.empty((world * rows[0], Sq, D), dtype=flat_local.dtype, device=flat_local.device)
        dist.all_gather_into_tensor(full, flat_local, group=group)
    else:  # ragged split: pad to the largest shard
        mx = max(rows)
        pad = torch.zeros((mx, Sq, D), dtype=flat_local.dtype, device=flat_local.device)
        pad[:flat_local.shape[0]] = flat_local
        buf = torch.empty((world * mx, Sq, D), dtype=flat_local.dtype, device=flat_local.device)
        dist.all_gather_into_tensor(buf, pad, group=group)
        full = torch.cat([buf[r * mx:r * mx + rows[r]] for r in range(world)], dim=0)
    return full.reshape(B, Hq, Sq, D)
