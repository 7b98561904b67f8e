"""Worker for tests/test_gpu_train_detr.py::test_syncbn_two_ranks_match_global_batch: launched by torch.distributed.run with two
ranks that SHARE GPU 0 (gloo backend moves the tiny statistics vectors through the host; RCCL refuses two ranks on one
device).  Each rank runs ResNet50-vd in SyncBN mode on its half of a 4-image batch; rank 0 also runs plain BN on the whole
batch and checks that features, running statistics and the rank-summed gradients coincide."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from focoos_amd.registry import ModelRegistry  # noqa: E402
from focoos_amd.synth import synth_image_structured, synth_state_dict  # noqa: E402
from focoos_amd.train_nn import ResNetVd, set_norm_mode  # noqa: E402


def rel(a, b):
    return float((a.float() - b.float()).norm() / (b.float().norm() + 1e-12))


def run(mode, imgs, bsd):
    net = set_norm_mode(ResNetVd(50), mode).to("cuda:0")
    net.load_state_dict(bsd, strict=True)
    outs = net(torch.from_numpy(np.stack(imgs)).to("cuda:0"))
    loss = sum((outs[k].float() ** 2).sum() for k in outs) * 0.5e-3
    loss.backward()
    torch.cuda.synchronize()
    return net, outs


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    assert world == 2
    torch.cuda.set_device(0)
    cfg = ModelRegistry.get_model_info("fai-detr-l-coco")["config"]
    sd = synth_state_dict(cfg, 12)
    pre = "pixel_decoder.backbone."
    bsd = {k[len(pre):]: v for k, v in sd.items() if k.startswith(pre)}
    imgs = [synth_image_structured(50 + i, 128, 160) for i in range(4)]
    net, outs = run("SyncBN", imgs[rank * 2:(rank + 1) * 2], bsd)
    grads = {n: p.grad.detach().clone() for n, p in net.named_parameters()}
    for g in grads.values():
        dist.all_reduce(g)           # the data-parallel SUM (TrainStep's reducer divides by world afterwards)
    stats = {k: v.detach().clone() for k, v in net.state_dict().items() if "running" in k}
    other = {k: v.clone() for k, v in stats.items()}
    for v in other.values():
        dist.broadcast(v, 1)
    ok = True
    if rank == 0:
        ref, ref_outs = run("BN", imgs, bsd)
        worst_f = max(rel(outs[k], ref_outs[k][:2]) for k in outs)
        worst_s = max(rel(stats[k], ref.state_dict()[k]) for k in stats)
        worst_x = max(rel(stats[k], other[k]) for k in stats)
        errs = sorted(((rel(grads[n], p.grad), n) for n, p in ref.named_parameters()), reverse=True)
        print(f"SYNCBN features {worst_f:.2e} running-stats {worst_s:.2e} cross-rank-stats {worst_x:.2e} grads worst {errs[0][0]:.2e} ({errs[0][1]}) "
              f"median {errs[len(errs) // 2][0]:.2e}", flush=True)
        # identical mathematics; differences = fp32 atomic summation order (statistics) amplified through 50 bf16 layers
        ok = worst_f <= 2e-2 and worst_s <= 1e-3 and worst_x <= 1e-6 and errs[0][0] <= 0.2 and errs[len(errs) // 2][0] <= 0.06
    flag = torch.tensor([1 if ok else 0])
    dist.broadcast(flag, 0)
    dist.destroy_process_group()
    sys.exit(0 if int(flag) == 1 else 1)


if __name__ == "__main__":
    main()
