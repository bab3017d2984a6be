"""One rank of tests/test_gpu_dist.py::test_peer_exchange_two_processes_on_one_gpu:  python peer_exchange_worker.py RANK WORLD
(MASTER_ADDR / MASTER_PORT in the environment; every rank uses cuda:0 -- the box has one GPU)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "aule-attention_amd"))
import torch
import torch.distributed as dist


def main(rank, world):
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import aule
    from aule import dist as adist
    ok = True
    # 2 + 2, ragged 2 + 1, (batch, kv-head) units; then two more sizes and the first one again: six distinct buffer sizes walk the
    # exchange cache past its four keys (aule/dist.py: the least recently used pair is closed, on every rank at the same call)
    for (B, Hq, Hkv, S) in ((4, 8, 2, 512), (3, 8, 8, 320), (1, 8, 2, 256), (2, 8, 2, 128), (2, 4, 4, 192), (4, 8, 2, 512)):
        g = torch.Generator(device="cuda").manual_seed(5)
        q = torch.randn(B, Hq, S, 128, device="cuda", dtype=torch.bfloat16, generator=g)
        k = torch.randn(B, Hkv, S, 128, device="cuda", dtype=torch.bfloat16, generator=g)
        v = torch.randn(B, Hkv, S, 128, device="cuda", dtype=torch.bfloat16, generator=g)
        ref = aule.flash_attention(q, k, v, causal=True)
        for chunks in (1, 3):
            for _ in range(2):
                full = adist.flash_attention_sharded(q, k, v, causal=True, chunks=chunks, transport="peer")
                ok = ok and bool(torch.equal(full, ref))
    Bl = 2
    g = torch.Generator(device="cuda").manual_seed(11 + rank)
    q = torch.randn(Bl, 8, 384, 64, device="cuda", dtype=torch.float16, generator=g)
    k = torch.randn(Bl, 8, 384, 64, device="cuda", dtype=torch.float16, generator=g)
    v = torch.randn(Bl, 8, 384, 64, device="cuda", dtype=torch.float16, generator=g)
    mine = aule.flash_attention(q, k, v, causal=False)
    full = adist.attention_and_gather(q, k, v, causal=False, chunks=2, transport="peer")
    ok = ok and bool(torch.equal(full[rank * Bl:(rank + 1) * Bl], mine)) and tuple(full.shape) == (world * Bl, 8, 384, 64)
    sums = [None] * world
    dist.all_gather_object(sums, float(full.float().sum().item()))
    ok = ok and len(set(sums)) == 1          # every rank holds the same gathered tensor
    ok = ok and len(adist._peer_cache) <= adist._PEER_CACHE_KEYS
    torch.cuda.synchronize()
    adist.release_peer_buffers()
    dist.barrier()
    dist.destroy_process_group()
    print("RANK_OK" if ok else "RANK_BAD", rank, flush=True)


if __name__ == "__main__":
    main(int(sys.argv[1]), int(sys.argv[2]))
