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
    red.launch()
    red.wait()
    other = torch.randn(10_000, generator=torch.Generator().manual_seed(100 + (1 - rank)))
    q.put((rank, bool(torch.allclose(flat, (mine + other) / 2, atol=1e-6))))
    dist.destroy_process_group()


def _bf16_worker(rank, world, port, q):
    """FX_DP_BF16 / bf16=True: buckets travel as bfloat16 (half the xGMI bytes); the averaged gradient equals the fp32 average to bf16
    rounding of the two summands and of their sum (VERDICT r3 next #9c)."""
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from focoos_amd.train import BucketedGradAllReduce

    flat = torch.randn(10_000, generator=torch.Generator().manual_seed(200 + rank)) * 3.0
    mine = flat.clone()
    red = BucketedGradAllReduce(flat, bucket_bytes=4096 * 4, segments=[(0, 6000), (6000, 10_000)], bf16=True)
    red.launch_segment(1)
    staged_dtype = red._staged[0][2].dtype
    red.launch()
    red.wait()
    other = torch.randn(10_000, generator=torch.Generator().manual_seed(200 + (1 - rank))) * 3.0
    want = (mine + other) / 2
    exact = (mine.bfloat16().float() + other.bfloat16().float()).bfloat16().float() / 2     # what a bf16 collective computes
    err = float((flat - want).abs().max() / want.abs().max())
    q.put((rank, staged_dtype == torch.bfloat16, bool(torch.equal(flat, exact)), err < 8e-3, not red._staged))
    dist.destroy_process_group()


def test_bf16_buckets_two_ranks_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_bf16_worker, args=(r, 2, 29715, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(60)
    assert res == [(0, True, True, True, True), (1, True, True, True, True)], res


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


def _overlap_worker(rank, world, port, q):
    """A three-segment network [backbone | encoder | head] whose gradients land in one flat buffer, reduced the way TrainStep does it
    (focoos_amd/train_detr.py): hooks on the separating activations start a segment's buckets while backward is still running."""
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from focoos_amd.train import BucketedGradAllReduce, notify_when_all_grads

    torch.manual_seed(0)   # same parameters on both ranks
    mods = [torch.nn.Linear(64, 64), torch.nn.Linear(64, 64), torch.nn.Linear(64, 8)]
    params = [p for m in mods for p in m.parameters()]
    flat = torch.zeros(sum(p.numel() for p in params))
    o, bounds = 0, []
    for m in mods:
        bounds.append(o)
        for p in m.parameters():
            p.grad = flat[o:o + p.numel()].view_as(p)   # gradients accumulate straight into the flat buffer
            o += p.numel()
    segments = [(bounds[0], bounds[1]), (bounds[1], bounds[2]), (bounds[2], o)]
    red = BucketedGradAllReduce(flat, bucket_bytes=1024 * 4, segments=segments)
    assert all(lo <= s and s + n <= hi for (lo, hi), sb in zip(segments, red.seg_buckets) for s, n in sb)   # no bucket straddles a segment
    events = []
    orig = red.launch_segment

    def traced(i):
        if not red.launched[i]:
            events.append(("launch", i, tuple(bool((flat[a:b] != 0).any()) for a, b in segments)))
        orig(i)

    red.launch_segment = traced
    x = torch.randn(16, 64, generator=torch.Generator().manual_seed(10 + rank))
    f = mods[0](x).relu()
    notify_when_all_grads([f], lambda name: red.launch_segment(1), "encoder")
    e = mods[1](f).relu()
    notify_when_all_grads([e], lambda name: red.launch_segment(2), "head")
    loss = mods[2](e).square().mean()
    loss.backward()
    red.launch()
    red.wait()
    # reference: the same network, both ranks' inputs, plain autograd
    torch.manual_seed(0)
    ref = [torch.nn.Linear(64, 64), torch.nn.Linear(64, 64), torch.nn.Linear(64, 8)]
    tot = None
    for r in range(world):
        xr = torch.randn(16, 64, generator=torch.Generator().manual_seed(10 + r))
        l = ref[2](ref[1](ref[0](xr).relu()).relu()).square().mean() / world
        l.backward()
    want = torch.cat([p.grad.flatten() for m in ref for p in m.parameters()])
    ok = bool(torch.allclose(flat, want, atol=1e-6))
    q.put((rank, ok, [(k, i) for k, i, _ in events], [st for _, _, st in events], red.log))
    dist.destroy_process_group()


def test_allreduce_overlaps_backward_two_ranks_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_overlap_worker, args=(r, 2, 29713, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(60)
    for rank, ok, order, state, log in res:
        assert ok, "averaged gradients differ from the single-process gradient of the global batch"
        assert order == [("launch", 2), ("launch", 1), ("launch", 0)]            # reverse parameter order, as backward finalises them
        # when the head's buckets were launched the backbone (and encoder) gradients did not exist yet: backward was still running
        assert state[0] == (False, False, True) and state[1] == (False, True, True)
        assert log == [("segment", 2), ("segment", 1), ("backward_end", -1), ("segment", 0)]   # two of three segments in flight before backward ended


def _status_worker(rank, world, port, q):
    """raise_if_infeasible(all_ranks=True): the Hungarian status word is a BITFIELD (bit 0 infeasible, bit 1 invalid entries) - rank 0 sets
    bit 0, rank 1 bit 1; after the bitwise-OR all-reduce both ranks hold 3, raise the same error in the same step and clear the word.
    A second round with nothing pending on rank 0 and bit 0 on rank 1: both raise the 'infeasible' message."""
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from focoos_amd.criterion import lsa_status, raise_if_infeasible

    out = []
    st = lsa_status("cpu")
    st.fill_(1 if rank == 0 else 2)
    try:
        raise_if_infeasible("cpu", all_ranks=True)
        out.append(None)
    except ValueError as e:
        out.append(str(e))
    out.append(int(st.item()))
    if rank == 1:
        st.fill_(1)
    try:
        raise_if_infeasible("cpu", all_ranks=True)
        out.append(None)
    except ValueError as e:
        out.append(str(e))
    raise_if_infeasible("cpu", all_ranks=True)   # nothing pending anywhere: no error on either rank
    q.put((rank, tuple(out)))
    dist.destroy_process_group()


def test_status_word_is_or_reduced_across_ranks():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_status_worker, args=(r, 2, 29719, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(60)
    want = ("matrix contains invalid numeric entries", 0, "cost matrix is infeasible")
    assert res == [(0, want), (1, want)], res
