"""Worker for tests/test_gpu_train_detr.py::test_syncbn_two_ranks_match_global_batch: launched by torch.distributed.run with two
ranks that SHARE GPU 0 (gloo backend moves the tiny statistics vectors through the host; RCCL refuses two ranks on one
device).  Each rank runs two ResNet bottlenecks (7 BatchNorm layers) in SyncBN mode on its half of a 4-image batch; rank 0
also runs plain BN on the whole batch and checks that features, input gradients, running statistics and the rank-summed
parameter gradients coincide - and that BN on the half batch alone (the control) does NOT."""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from focoos_amd.train_nn import set_norm_mode  # noqa: E402


def rel(a, b):
    return float((a.float() - b.float()).norm() / (b.float().norm() + 1e-12))


def build(mode, sd):
    from focoos_amd import _lib
    from focoos_amd.train_nn import BottleNeck, _Blocks

    lib = _lib.load()
    net = set_norm_mode(_Blocks([BottleNeck(lib, 64, 32, 1, False, True), BottleNeck(lib, 128, 32, 1, True, True)]), mode)
    if sd is None:
        g = torch.Generator().manual_seed(5)
        with torch.no_grad():
            for n, p in net.named_parameters():
                if n.endswith("conv.weight"):
                    p.copy_(torch.randn(p.shape, generator=g) / (p.shape[1] * p.shape[2] * p.shape[3]) ** 0.5)
                elif n.endswith("norm.weight"):
                    p.copy_(torch.rand(p.shape, generator=g) + 0.5)
                else:
                    p.copy_(torch.randn(p.shape, generator=g) * 0.3)
    else:
        net.load_state_dict(sd)
    return net.to("cuda:0")


def run(mode, x, cot, sd):
    net = build(mode, sd)
    xd = x.to("cuda:0").requires_grad_(True)
    y = net(xd)
    y.backward(cot.to("cuda:0"))
    torch.cuda.synchronize()
    return net, y.detach(), xd.grad


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    assert world == 2
    torch.cuda.set_device(0)
    g = torch.Generator().manual_seed(9)
    x = (torch.randn(4, 24, 28, 64, generator=g) + torch.arange(4).view(4, 1, 1, 1) * 0.5).clamp_min(0).bfloat16()   # per-image offset: the
    cot = torch.randn(4, 24, 28, 128, generator=g).bfloat16()                                                       # halves have different statistics
    sd0 = {k: v.cpu().clone() for k, v in build("BN", None).state_dict().items()}
    sl = slice(rank * 2, rank * 2 + 2)
    net, y, dx = run("SyncBN", x[sl], cot[sl], sd0)
    grads = {n: p.grad.detach().clone() for n, p in net.named_parameters()}
    for gr in grads.values():
        dist.all_reduce(gr)           # the data-parallel SUM (TrainStep's reducer divides by world afterwards)
    stats = {k: v.detach().clone() for k, v in net.state_dict().items() if "running" in k}
    other = {k: v.clone() for k, v in stats.items()}
    for v in other.values():
        dist.broadcast(v, 1)
    ok = True
    if rank == 0:
        ref, ref_y, ref_dx = run("BN", x, cot, sd0)                 # the whole batch in one process
        loc, loc_y, _ = run("BN", x[sl], cot[sl], sd0)              # control: local statistics only (what SyncBN must NOT equal)
        f, fx = rel(y, ref_y[:2]), rel(dx, ref_dx[:2])
        worst_s = max(rel(stats[k], ref.state_dict()[k]) for k in stats)
        worst_x = max(rel(stats[k], other[k]) for k in stats)
        errs = sorted(((rel(grads[n], p.grad), n) for n, p in ref.named_parameters()), reverse=True)
        control = rel(loc_y, ref_y[:2])
        print(f"SYNCBN features {f:.2e} dx {fx:.2e} running-stats {worst_s:.2e} cross-rank-stats {worst_x:.2e} grads worst {errs[0][0]:.2e} "
              f"({errs[0][1]}) median {errs[len(errs) // 2][0]:.2e}; control (local statistics) features {control:.2e}", flush=True)
        # identical mathematics; what is left is fp32 summation order of the statistics decorrelating bf16 roundings downstream
        ok = f <= 1e-2 and fx <= 0.1 and worst_s <= 1e-3 and worst_x <= 1e-6 and errs[0][0] <= 0.1 and control >= 5 * f
    flag = torch.tensor([1 if ok else 0])
    dist.broadcast(flag, 0)
    dist.destroy_process_group()
    sys.exit(0 if int(flag) == 1 else 1)


if __name__ == "__main__":
    main()
