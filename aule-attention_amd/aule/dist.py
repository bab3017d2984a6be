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
(contiguous views, each piece written by the kernel straight into its place in the gathered tensor -- no change to the
kernels: a piece is an ordinary batch of independent heads), and the
gather of piece i is launched asynchronously (RCCL runs it on its own stream, ordered behind the kernel that produced
it) while piece i+1 computes; (2) `transport="p2p"` posts the direct sends / receives of a piece to all peers as one
batch (torch.distributed.batch_isend_irecv -> one RCCL group: all links busy, and shards of different sizes need no
padding), `transport="allgather"` posts one all_gather_into_tensor per piece into a contiguous staging tensor that one
strided copy moves to its place (the collective's output is rank-major, the gathered tensor shard-major), and
`transport="peer"` leaves RCCL out of the data path altogether: the gathered tensor lives in a buffer every peer has mapped
(include/aule.h aule_peer_*: hipMalloc + hipIpcGetMemHandle here, hipIpcOpenMemHandle there), and a finished piece is
copied straight into the same place of every peer's buffer by one hipMemcpyAsync per peer on that peer's own stream -- seven
independent xGMI links, no algorithm choice, no protocol thresholds (class PeerExchange below).  Expected
end-to-end cost of the gather at config 4: ~0.4 ms exposed (the last piece) instead of 3.5 ms (p2p) / 24.5 ms (ring).
Every rank ends with the full [B, Hq, Sq, D] tensor; pass gather=False to keep the shard (what a data-parallel model does:
it never needs the other ranks' attention outputs).  The backward needs no collective at all: dQ, dK, dV are per unit.
"""
import os
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


def shard_layout(batch: int, heads_q: int, heads_kv: int, world: int):
    """Where every rank's shard sits in the flattened [B*Hq, Sq, D] output: (mode, n_lead, row0, per_lead, step) -- n_lead[r] =
    extent of rank r's shard along its leading axis (batch items, or query heads of the flattened unit axis), row0[r] = its first
    output row, per_lead = output rows per leading index, step = granularity of a piece (a query group stays whole)."""
    g = heads_q // heads_kv
    mode, ranges = shard_plan(batch, heads_kv, world)
    per_lead = heads_q if mode == "batch" else 1
    units_to_lead = 1 if mode == "batch" else g
    n_lead = [(e - s) * units_to_lead for s, e in ranges]
    row0 = [s * units_to_lead * per_lead for s, _ in ranges]
    return mode, n_lead, row0, per_lead, (1 if mode == "batch" else g)


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
                            transport: str = "auto", window_size: int = -1):
    """Every rank holds the same full q, k, v (or at least its own shard's rows); each computes its share with `attn_fn`
    (default aule.flash_attention) in `chunks` pieces and, if `gather`, every rank receives the full output: the exchange of
    piece i overlaps the computation of piece i+1 (module docstring).  transport: "allgather", "p2p", "peer" (direct copies into the peers' mapped buffers; the
    result then lives in a cached exchange buffer and stays valid until the second-next "peer" exchange of the same size, or until more
    than AULE_PEER_CACHE_KEYS = 4 other exchange sizes evict its buffer: see _peer_exchange; .clone() a result that must outlive that) or
    "auto" (p2p when the shards differ in size, all-gather otherwise).  Returns the full [B,Hq,Sq,D] output (gather=True) or this rank's
    shard.  Inference path (no autograd through the collective).  chunks=1, transport="allgather" is the single blocking
    collective of round 1.  window_size > 0: the sliding window of aule.flash_attention (round 6; every shard is a set of whole heads: the window needs
    nothing from another rank); a caller-supplied attn_fn receives it as `window_size=` only when it is set."""
    import torch
    import torch.distributed as dist
    into = _into_ok(q, k, v, causal, attn_fn)
    wkw = {"window_size": int(window_size)} if window_size is not None and window_size > 0 else {}
    if attn_fn is None:
        from . import flash_attention as attn_fn
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    B, Hq, Sq, D = q.shape
    Hkv = k.shape[1]
    g = Hq // Hkv
    mode, qs, ks, vs = local_shard(q, k, v, rank, world)
    _, n_lead, row0, per_lead, _ = shard_layout(B, Hq, Hkv, world)
    lead = 0 if mode == "batch" else 1                      # batch items, or heads of the single flattened batch item

    def run(sl, out=None):
        qq = qs[sl] if lead == 0 else qs[:, sl]
        if lead == 0:
            kk, vv = ks[sl], vs[sl]
        else:   # heads of one flattened batch item: the KV heads of the same units
            kk, vv = ks[:, sl.start // g:(sl.stop + g - 1) // g], vs[:, sl.start // g:(sl.stop + g - 1) // g]
        if out is not None:
            return _attn_into(qq, kk, vv, causal, scale, out.view(qq.shape), window_size)
        return attn_fn(qq.contiguous(), kk.contiguous(), vv.contiguous(), causal=causal, scale=scale, **wkw)

    if not gather:
        if n_lead[rank] == 0:
            return qs.new_empty(qs.shape)
        return run(slice(0, n_lead[rank]))
    full = _compute_and_exchange(run, n_lead, row0, per_lead, g if lead == 1 else 1, rank, world, (B * Hq, Sq, D),
                                 q.dtype, q.device, chunks, transport, group, into=into)
    return full.reshape(B, Hq, Sq, D)


def attention_and_gather(q, k, v, causal: bool = True, scale: Optional[float] = None, group=None,
                         attn_fn: Optional[Callable] = None, chunks: int = 4, transport: str = "allgather", window_size: int = -1):
    """The same exchange for ranks that hold ONLY their own shard (equal shards along the batch axis: what a data-parallel
    caller and bench.py have): q [Bl,Hq,Sq,D], k / v [Bl,Hkv,Sk,D] local; returns [world*Bl,Hq,Sq,D] on every rank,
    rank r's rows at [r*Bl, (r+1)*Bl)."""
    import torch.distributed as dist
    into = _into_ok(q, k, v, causal, attn_fn)
    wkw = {"window_size": int(window_size)} if window_size is not None and window_size > 0 else {}
    if attn_fn is None:
        from . import flash_attention as attn_fn
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    Bl, Hq, Sq, D = q.shape

    def run(sl, out=None):
        if out is not None:
            return _attn_into(q[sl], k[sl], v[sl], causal, scale, out.view(q[sl].shape), window_size)
        return attn_fn(q[sl].contiguous(), k[sl].contiguous(), v[sl].contiguous(), causal=causal, scale=scale, **wkw)

    full = _compute_and_exchange(run, [Bl] * world, [r * Bl * Hq for r in range(world)], Hq, 1, rank, world,
                                 (world * Bl * Hq, Sq, D), q.dtype, q.device, chunks, transport, group, into=into)
    return full.reshape(world * Bl, Hq, Sq, D)


def _into_ok(q, k, v, causal, attn_fn):
    """Can the default attention write its result straight into a view of the gathered tensor?  Only for what fwd_raw takes
    as it is: device tensors, q / k / v of ONE dtype the kernels run natively, a head_dim they take unpadded, and a
    bottom-right mask with seq_len_k >= seq_len_q.  The reference's shape rules are checked here (ValueError, as from
    aule.flash_attention) because the in-place route does not pass through it; everything else -- mixed dtypes, padded head
    dims, a caller-supplied attn_fn -- takes the copying route through aule.flash_attention, which casts and checks itself."""
    from . import _validate
    _validate(q, k, v)
    if attn_fn is not None or not all(getattr(t, "is_cuda", False) for t in (q, k, v)):
        return False
    from . import _torch as at
    if q.dtype not in at._DTYPES or k.dtype != q.dtype or v.dtype != q.dtype:
        return False
    if q.shape[-1] not in at.SUPPORTED_HEAD_DIMS:
        return False
    if at.causal_code(causal) == 2 and k.shape[2] < q.shape[2]:
        return False          # (aule.flash_attention raises the ValueError on the copying route)
    return True


def _attn_into(q, k, v, causal, scale, out, window_size=-1):
    """aule.flash_attention (inference form: no autograd node, no LSE) with the result written to `out`."""
    import math
    from . import _torch as at
    sc = float(scale) if scale is not None else 1.0 / math.sqrt(q.shape[-1])
    at.fwd_raw(q.contiguous(), k.contiguous(), v.contiguous(), at.causal_code(causal), sc, want_lse=False, out=out,
               window=int(window_size) if window_size is not None and window_size > 0 else -1)
    return out


def peer_copy_plan(n_lead, row0, per_lead, step, rank, chunks, row_bytes):
    """The pieces of rank `rank`'s shard as byte ranges of the gathered [rows, Sq, D] tensor: [(lead_a, lead_b, byte offset,
    bytes)].  Every rank's buffer has the same layout, so a piece goes to the SAME offset of every peer's buffer -- the
    offset arithmetic of transport="peer" in one place (tests/test_dist_gloo.py checks that the ranks' plans tile the tensor
    exactly once)."""
    out = []
    for a, b in chunk_ranges(n_lead[rank] // step, chunks):
        a, b = a * step, b * step
        out.append((a, b, (row0[rank] + a * per_lead) * row_bytes, (b - a) * per_lead * row_bytes))
    return out


class PeerExchange:
    """The gathered tensor of transport="peer": a device buffer of this rank that every peer has mapped, plus this rank's
    mappings of the peers' buffers and one stream per peer.  Built collectively (every rank of `group` must construct it with
    the same nbytes); cached by _compute_and_exchange per (group, bytes, device), two buffers alternating, so a result stays
    valid until the second-next exchange of the same size on the same group, or until its cache entry is evicted by exchanges of more
    than AULE_PEER_CACHE_KEYS (default 4) other sizes / groups, whichever comes first (_peer_exchange below): clone what must live longer."""

    def __init__(self, nbytes, device, group):
        import ctypes
        import torch
        import torch.distributed as dist
        from . import _capi
        self.lib = _capi.load()
        self.nbytes, self.device, self.group = int(nbytes), device, group
        self.dev = device.index if device.index is not None else torch.cuda.current_device()
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self.ptr = ctypes.c_void_p()
        h = _capi.IpcHandle()
        err = None
        rc = self.lib.aule_peer_alloc(self.dev, self.nbytes, ctypes.byref(self.ptr), ctypes.byref(h))
        if rc != 0:
            err = self.lib.aule_get_error().decode()
            self.ptr = ctypes.c_void_p()
        got = [None] * self.world
        dist.all_gather_object(got, (err, bytes(h.bytes)), group=group)
        self.remote = [None] * self.world
        if err is None and all(e is None for e, _ in got):
            for r, (_, hb) in enumerate(got):
                if r == self.rank:
                    continue
                hh = _capi.IpcHandle()
                ctypes.memmove(hh.bytes, hb, 64)
                p = ctypes.c_void_p()
                if self.lib.aule_peer_open(self.dev, ctypes.byref(hh), ctypes.byref(p)) != 0:
                    err = self.lib.aule_get_error().decode()
                    break
                self.remote[r] = p
        # every rank learns whether every rank is set up (a rank that failed must not leave the others waiting in the exchange)
        oks = [None] * self.world
        dist.all_gather_object(oks, err if err is not None else next((e for e, _ in got if e is not None), None), group=group)
        bad = next((e for e in oks if e is not None), None)
        if bad is not None:
            self.close()
            raise _capi.AuleError(f"peer exchange set-up failed on some rank: {bad}")
        self.streams = [torch.cuda.Stream(device=device) if r != self.rank else None for r in range(self.world)]
        self._arr = _RawDeviceBytes(self.ptr.value, self.nbytes)
        self.bytes_view = torch.as_tensor(self._arr, device=device)    # uint8 [nbytes], zero-copy

    def tensor(self, shape, dtype):
        return self.bytes_view.view(dtype).reshape(shape)

    def send(self, offset, nbytes, event):
        """Copy [offset, offset + nbytes) of this rank's buffer to the same place of every peer's, each on its own stream,
        behind `event` (recorded on the stream that produced the bytes)."""
        import ctypes
        from . import _capi
        for r in range(self.world):
            if r == self.rank:
                continue
            st = self.streams[r]
            st.wait_event(event)
            _capi.check(self.lib.aule_peer_copy_async(self.dev, ctypes.c_void_p(self.remote[r].value + offset),
                                                      ctypes.c_void_p(self.ptr.value + offset), nbytes,
                                                      ctypes.c_void_p(st.cuda_stream)), "aule_peer_copy_async")

    def finish(self):
        """Returns when every rank's copies have landed everywhere: each rank waits for its own outgoing copies, then the ranks
        meet (a one-element all-reduce: control only, no payload)."""
        import torch
        import torch.distributed as dist
        for st in self.streams:
            if st is not None:
                st.synchronize()
        nccl = dist.get_backend(self.group) == "nccl"
        flag = torch.zeros(1, device=self.device if nccl else "cpu")
        dist.all_reduce(flag, group=self.group)
        if nccl:      # the all-reduce is only stream-ordered there: make the host (and so every stream it launches on) wait
            torch.cuda.current_stream(self.device).synchronize()

    def close(self):
        for r, p in enumerate(getattr(self, "remote", [])):
            if p is not None:
                self.lib.aule_peer_close(self.dev, p)
                self.remote[r] = None
        if getattr(self, "ptr", None) is not None and self.ptr.value:
            self.lib.aule_peer_free(self.dev, self.ptr)
            self.ptr.value = None


class _RawDeviceBytes:
    """A device allocation as a uint8 array for torch.as_tensor (CUDA array interface v3)."""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (int(nbytes),), "typestr": "|u1", "data": (int(ptr), False), "version": 3,
                                         "strides": None}


_peer_cache = {}          # insertion-ordered: the least recently used key first
# distinct (group, bytes, device) triples kept; the oldest pair of buffers is closed (and FREED) beyond that.  A workload whose
# exchanges come in more sizes than this pays a collective IPC set-up per call: raise AULE_PEER_CACHE_KEYS for it.
_PEER_CACHE_KEYS = max(1, int(os.environ.get("AULE_PEER_CACHE_KEYS", "4")))


def _peer_exchange(nbytes, device, group):
    """Two alternating buffers per (group, bytes, device).  The entry holds the group object itself, so its id() cannot be
    reused by another group while the entry lives; at most _PEER_CACHE_KEYS entries are kept (every rank runs the same
    sequence of exchanges, so every rank evicts the same entry at the same call -- closing stays collective).

    LIFETIME of a result: a tensor returned by a transport="peer" gather aliases one of these buffers (zero-copy).  It stays
    valid until (a) the second-next peer exchange of the SAME (group, bytes, device) overwrites it, or (b) its entry is evicted -- the
    exchange that brings the number of distinct keys above _PEER_CACHE_KEYS (default 4, AULE_PEER_CACHE_KEYS) closes and frees the least
    recently used pair -- or (c) release_peer_buffers().  Callers that keep a gathered tensor across later exchanges must .clone() it."""
    key = (id(group) if group is not None else 0, int(nbytes), str(device))
    ent = _peer_cache.pop(key, None)
    if ent is None:
        while len(_peer_cache) >= _PEER_CACHE_KEYS:
            old = _peer_cache.pop(next(iter(_peer_cache)))
            for b in old["bufs"]:
                b.close()
        ent = {"bufs": [PeerExchange(nbytes, device, group), PeerExchange(nbytes, device, group)], "n": 0, "group": group}
    _peer_cache[key] = ent      # most recently used: last
    ent["n"] += 1
    return ent["bufs"][ent["n"] & 1]


def release_peer_buffers():
    """Close every cached peer-exchange buffer (collective in spirit: call it on every rank before the process group goes).
    Tensors returned by earlier exchanges alias those buffers and must not be used afterwards."""
    for ent in _peer_cache.values():
        for b in ent["bufs"]:
            b.close()
    _peer_cache.clear()


def _compute_and_exchange(run, n_lead, row0, per_lead, step, rank, world, shape, dtype, device, chunks, transport, group,
                          into=False):
    """run(slice over the shard's leading axis, out) -> that piece's output (written to `out` when into=True and out is given);
    n_lead[r] = leading extent of rank r's shard, row0[r] = its first row in the flattened [rows, Sq, D] result, per_lead = rows
    per leading index, step = granularity of a piece (a query group stays whole).  Computes this rank's pieces and exchanges
    them, piece i's exchange overlapping piece i+1's kernels."""
    import torch
    import torch.distributed as dist
    Sq, D = shape[1], shape[2]
    if transport == "auto":
        transport = "allgather" if len(set(n_lead)) == 1 else "p2p"
    if transport not in ("allgather", "p2p", "peer"):
        raise ValueError(f"transport must be 'auto', 'allgather', 'p2p' or 'peer', got {transport!r}")
    if transport == "allgather" and len(set(n_lead)) != 1:
        raise ValueError("transport='allgather' needs equal shards; use 'p2p', 'peer' (or 'auto') for a ragged split")

    def piece_of(full, a, b):
        """piece [a, b) of this rank's shard, computed into its place in `full` when the attention can do that"""
        view = full[row0[rank] + a * per_lead: row0[rank] + b * per_lead]
        if into:
            run(slice(a, b), view)
        else:
            view.copy_(run(slice(a, b), None).reshape(-1, Sq, D))
        return view

    if transport == "peer":
        if device.type != "cuda":
            raise ValueError("transport='peer' exchanges device buffers; these tensors are on " + str(device))
        row_bytes = Sq * D * torch.empty((), dtype=dtype).element_size()
        px = _peer_exchange(shape[0] * row_bytes, device, group)
        full = px.tensor(shape, dtype)
        cur = torch.cuda.current_stream(device)
        for a, b, off, nb in peer_copy_plan(n_lead, row0, per_lead, step, rank, chunks, row_bytes):
            piece_of(full, a, b)
            ev = torch.cuda.Event()
            ev.record(cur)
            px.send(off, nb, ev)
        px.finish()
        return full

    full = torch.empty(shape, dtype=dtype, device=device)
    works = []
    if transport == "allgather":
        # equal shards: full is [world, shard rows, Sq, D].  Piece i of every rank is gathered into ONE contiguous
        # [world, piece rows, Sq, D] staging tensor (all_gather_into_tensor: no per-rank temporaries inside the process group,
        # which is what a list of strided views costs) and lands in `full` with one strided copy behind the collective.
        shard_rows = n_lead[rank] * per_lead
        full4 = full.view(world, shard_rows, Sq, D)
        staged = []
        for a, b in [(a * step, b * step) for a, b in chunk_ranges(n_lead[rank] // step, chunks)]:
            piece = piece_of(full, a, b)
            stage = torch.empty((world * (b - a) * per_lead, Sq, D), dtype=dtype, device=device)   # rank-major concatenation
            staged.append((dist.all_gather_into_tensor(stage, piece, group=group, async_op=True), stage, a * per_lead, b * per_lead))
        for w, stage, ra, rb in staged:
            w.wait()
            full4[:, ra:rb].copy_(stage.view(world, rb - ra, Sq, D))
    else:
        # every rank cuts ITS shard into the same number of pieces (possibly of different sizes); piece i of every rank
        # is exchanged in one batch of sends / receives
        npiece = max(1, min(int(chunks), max(1, min(n // step for n in n_lead if n > 0) if any(n_lead) else 1)))
        cuts = [[(a * step, b * step) for a, b in (partition(n // step, npiece) if n > 0 else [(0, 0)] * npiece)]
                for n in n_lead]
        for i in range(npiece):
            a, b = cuts[rank][i]
            piece = piece_of(full, a, b) if b > a else None      # in place: no second copy of the local piece
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
