"""Data-parallel gradient averaging (BucketedGradAllReduce) with 2 gloo processes on CPU: the collective wiring the
reference gets from DDP (focoos/utils/distributed/dist.py:138-157), here explicit buckets + async all-reduce."""
import os

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from focoos_amd.train import BucketedGradAllReduce

    g = torch.Generator().manual_seed(100 + rank)
    flat = torch.randn(10_000, generator=g)
    mine = flat.clone()
    red = BucketedGradAllReduce(flat, bucket_bytes=4096 * 4)   # 3 buckets (4096, 4096, 1808)
    assert len(red.buckets) == 3
    red.launch(2, 3)   # reverse order, as gradients become ready during backward
    red.launch(0, 2)
    red.wait()
    other = torch.randn(10_000, generator=torch.Generator().manual_seed(100 + (1 - rank)))
    q.put((rank, bool(torch.allclose(flat, (mine + other) / 2, atol=1e-6))))
    dist.destroy_process_group()


def test_bucketed_allreduce_two_ranks_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, 29711, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(60)
    assert res == [(0, True), (1, True)]
