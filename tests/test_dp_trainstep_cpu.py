"""The REAL data-parallel host logic of a training step on two gloo ranks (CPU), kernels stubbed (VERDICT r3 next #9a): the real
``FAIDetrTrainable`` parameter tree (501 tensors, the reference's names), the real ``TrainStep`` constructor and ``step()`` - flat
parameter / gradient layout, segment boundaries, bucket layout, rank-0 broadcast, the autograd hooks ``FAIDetrTrainable.forward``
installs on the activations that separate the segments, the ``num_boxes`` all-reduce of ``SetCriterionTrain.forward``, bucketed
asynchronous all-reduce started from those hooks, optimizer step - with every HIP launch replaced by a CPU stand-in that produces
rank-dependent gradients for EVERY parameter.  (What DistributedDataParallel does for the reference: utils/distributed/dist.py:138-157;
``num_boxes``: fai_detr/modelling.py:568-570.)  Round 3 drove only a toy 3-Linear network through the reducer - and missed that the
production constructor never recognised its three segments (the backbone is registered after pixel_decoder's own layers), so the
overlapped path was dead code."""
import os
from unittest import mock

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _scalar_of(module: torch.nn.Module, seed: int) -> torch.Tensor:
    """A differentiable scalar that depends on every trainable parameter of ``module`` (fixed random projections)."""
    g = torch.Generator().manual_seed(seed)
    tot = None
    for p in module.parameters():
        if p.requires_grad:
            t = (p * torch.randn(p.shape, generator=g)).sum() * (1.0 / max(p.numel(), 1) ** 0.5)
            tot = t if tot is None else tot + t
    return tot


def _worker(rank, world, port, q):
    try:
        _worker_body(rank, world, port, q)
    except Exception:   # a failing rank must not leave the parent waiting for the queue
        import traceback

        q.put({"rank": rank, "error": traceback.format_exc()})
        raise


def _worker_body(rank, world, port, q):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    from focoos_amd import criterion as crit_mod
    from focoos_amd import train_nn
    from focoos_amd.ports import DETRTargets
    from focoos_amd.registry import ModelRegistry

    cfg = ModelRegistry.get_model_info("fai-detr-l-obj365")["config"]
    with mock.patch("torch.cuda.is_available", return_value=True):   # the module tree is plain nn.Parameters; no kernel runs in a constructor
        from focoos_amd.train_detr import FAIDetrTrainable, TrainStep

        model = FAIDetrTrainable(cfg, norm="FrozenBN")
    torch.manual_seed(1000 + rank)                                   # DIFFERENT initial parameters per rank: the constructor must broadcast rank 0's
    with torch.no_grad():
        for p in model.parameters():
            p.copy_(torch.randn(p.shape) * 0.02)
    first = {n: p.detach().clone() for n, p in [(n, p) for n, p in model.named_parameters() if p.requires_grad][:3]}

    # ---- CPU stand-ins for the three sub-networks (every parameter of each takes part; inputs differ per rank)
    bb, pd, pr = model.pixel_decoder.backbone, model.pixel_decoder, model.head.predictor
    own_pd = torch.nn.ModuleList([m for n, m in pd.named_children() if n != "backbone"])

    def backbone_fwd(images):
        s = _scalar_of(bb, 1)
        x = images.float().mean(dim=(1, 2, 3)).view(-1, 1, 1, 1) / 255.0
        return {f"res{k}": x * (1.0 + 0.01 * s) * torch.ones(1, 4, 2, 2) * k for k in (3, 4, 5)}

    def encoder_fwd(feats):
        s = _scalar_of(own_pd, 2)
        return [f * (1.0 + 0.01 * s) for f in feats]

    def predictor_fwd(enc, forced_topk=None):
        s = _scalar_of(pr, 3)
        B = enc[0].shape[0]
        v = sum(e.mean(dim=(1, 2, 3)) for e in enc).view(B, 1, 1)
        logits = (v * (1.0 + 0.01 * s)).expand(B, 5, 7)
        boxes = torch.sigmoid(v * s * 0.01).expand(B, 5, 4)
        return {"pred_logits": logits, "pred_boxes": boxes, "aux_outputs": [{"pred_logits": logits * 0.5, "pred_boxes": boxes}]}

    bb.forward, pd.forward, pr.forward = backbone_fwd, encoder_fwd, predictor_fwd
    seen_num_boxes = []

    def one_set(out, tg, num_boxes, fixed=None):     # the criterion's forward (targets packing, num_boxes all-reduce, set loop) stays real
        seen_num_boxes.append(num_boxes)
        return {"loss_vfl": out["pred_logits"].square().mean() / num_boxes, "loss_bbox": out["pred_boxes"].mean() / num_boxes,
                "loss_giou": out["pred_boxes"].square().mean()}, (None, None)

    model.head.criterion._one_set = one_set
    model.head.criterion.matcher.match_packed_sets = lambda ll, bl, tg: [(None, None)] * len(ll)

    with mock.patch.object(crit_mod, "h2d_i32", lambda lst, dev: torch.tensor(lst, dtype=torch.int32)), \
            mock.patch.object(train_nn, "pin_stream", lambda dev, on: None), \
            mock.patch.object(train_nn.WeightPacker, "pack", lambda self, dev: 0):
        stepper = TrainStep(model, lr=0.1, check_every=0)
        # rank 0's parameters everywhere (DDP's construction-time broadcast)
        bcast_ok = all(torch.equal(stepper.opt.params[n], first[n]) == (rank == 0) for n in first)
        gathered = [torch.zeros(3) for _ in range(world)]
        dist.all_gather(gathered, torch.stack([stepper.opt.params[n].flatten()[0] for n in first]))
        bcast_ok = bcast_ok and all(torch.equal(g, gathered[0]) for g in gathered)
        # ---- layout: [backbone | pixel decoder | head], three segments, 64 MiB buckets that never straddle a segment
        names = [n for n, _ in stepper.named]
        segs = [0 if n.startswith("pixel_decoder.backbone.") else (1 if n.startswith("pixel_decoder.") else 2) for n in names]
        red = stepper.reducer
        numel = stepper.opt.flat_g.numel()
        layout = {"sorted": segs == sorted(segs), "segments": red.segments, "numel": numel, "n_buckets": len(red.buckets),
                  "bucket_bytes": [n * 4 for _, n in red.buckets],
                  "no_straddle": all(lo <= s and s + n <= hi for (lo, hi), sb in zip(red.segments, red.seg_buckets) for s, n in sb),
                  "plan_buckets": [4 * n for n in __import__("focoos_amd.train", fromlist=["dp_plan"]).dp_plan(cfg, "fai_detr", "FrozenBN", world, 2)["bucket_elements"]],
                  "first_encoder_param": names[segs.index(1)], "first_head_param": names[segs.index(2)], "hooks_on": model.grad_ready is not None}

        def adamw_stub(self):   # plain SGD on the averaged gradient: enough to see that every rank applies the same update
            self.step_count += 1
            self.flat_p.sub_(self.flat_g * 0.1)

        images = torch.full((2, 8, 8, 3), 60.0 + 40.0 * rank)
        targets = [DETRTargets(labels=torch.zeros(3 + 4 * rank, dtype=torch.int64), boxes=torch.rand(3 + 4 * rank, 4)),
                   DETRTargets(labels=torch.zeros(1, dtype=torch.int64), boxes=torch.rand(1, 4))]
        # local gradient of THIS rank without any collective (reference for the averaged result); num_boxes as the reduced value
        events = []
        orig_launch = red.launch_segment

        def traced(i):
            if not red.launched[i]:
                events.append((i, tuple(bool((stepper.opt.flat_g[a:b] != 0).any()) for a, b in red.segments)))
            orig_launch(i)

        red.launch_segment = traced
        # this rank's LOCAL gradient at the same parameters, no gradient collective (hooks off); the criterion's num_boxes all-reduce
        # takes place in it too, on both ranks
        hooks, model.grad_ready = model.grad_ready, None
        stepper.opt.zero_grad()
        sum(model(images, targets).values()).backward()
        local = stepper.opt.flat_g.clone()
        model.grad_ready = hooks
        del seen_num_boxes[:]
        with mock.patch.object(type(stepper.opt), "step", adamw_stub):
            p_before = stepper.opt.flat_p.clone()
            losses = stepper.step(images, targets)
        g_avg = stepper.opt.flat_g.clone()
        p_after = stepper.opt.flat_p.clone()
        # ---- the same step through the STAGED backward (round 6: what a captured step replays - three backward stages, segment k's
        # all-reduce enqueued between stage k and stage k+1, no autograd hooks): same parameters, same batch -> the same averaged gradient
        with torch.no_grad():
            stepper.opt.flat_p.copy_(p_before)
        stepper.staged = True
        hook_log = list(red.log)
        del red.log[:], stepper.stage_log[:]
        staged_events = []

        def traced2(i):
            if not red.launched[i]:
                staged_events.append((i, tuple(bool((stepper.opt.flat_g[a:b] != 0).any()) for a, b in red.segments)))
            orig_launch(i)

        red.launch_segment = traced2
        with mock.patch.object(type(stepper.opt), "step", adamw_stub):
            stepper.step(images, targets)
        g_staged = stepper.opt.flat_g.clone()
        staged = {"log": list(stepper.stage_log), "events": staged_events, "red_log": list(red.log),
                  "equal": bool(torch.equal(g_staged, g_avg)), "p_equal": bool(torch.equal(stepper.opt.flat_p, p_after))}
    # every rank must now hold the SAME gradient and the same parameters; and the gradient must be the mean of the two local ones
    both = [torch.zeros_like(g_avg) for _ in range(world)]
    dist.all_gather(both, g_avg)
    same_grad = bool(torch.equal(both[0], both[1]))
    loc_all = [torch.zeros_like(local) for _ in range(world)]
    dist.all_gather(loc_all, local)
    mean_ok = bool(torch.allclose(g_avg, (loc_all[0] + loc_all[1]) / world, rtol=1e-5, atol=1e-8))
    covered = float((g_avg != 0).float().mean())
    upd_ok = bool(torch.allclose(p_after, p_before - 0.1 * g_avg))
    q.put({"rank": rank, "bcast_ok": bcast_ok, "layout": layout, "events": events, "log": hook_log, "same_grad": same_grad, "mean_ok": mean_ok,
           "covered": covered, "upd_ok": upd_ok, "num_boxes": seen_num_boxes[:2], "n_losses": len(losses), "staged": staged})
    dist.destroy_process_group()


def test_trainstep_host_logic_two_ranks_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, 29741, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=600) for _ in range(2)), key=lambda d: d["rank"])
    for p in procs:
        p.join(120)
    for r in res:
        assert "error" not in r, r["error"]
        lay = r["layout"]
        assert r["bcast_ok"], "TrainStep must start every rank from rank 0's parameters"
        assert lay["sorted"] and len(lay["segments"]) == 3 and lay["hooks_on"], lay          # [backbone | encoder | head] recognised, hooks installed
        assert lay["segments"][0][0] == 0 and lay["segments"][2][1] == lay["numel"] == 43_361_847   # trainable parameters of fai-detr-l-obj365 (frozen BN affine excluded)
        assert lay["first_encoder_param"].startswith("pixel_decoder.input_proj.") and lay["first_head_param"].startswith("head.predictor.")
        assert lay["plan_buckets"] == lay["bucket_bytes"], "bench.py --train --dry-run (train.dp_plan) must describe the buckets TrainStep really cuts"
        assert lay["no_straddle"] and sum(lay["bucket_bytes"]) == 4 * lay["numel"] and max(lay["bucket_bytes"]) <= 64 << 20
        # all-reduce volume of a step = the fp32 gradient buffer (173.4 MB), in the few large buckets xGMI rings want
        assert 170e6 < sum(lay["bucket_bytes"]) < 176e6 and lay["n_buckets"] <= 6, lay["bucket_bytes"]
        # backward finalises head -> encoder -> backbone; each segment's buckets start while the earlier layers' gradients do not exist yet
        assert [i for i, _ in r["events"]] == [2, 1, 0], r["events"]
        assert r["events"][0][1] == (False, False, True) and r["events"][1][1] == (False, True, True), r["events"]
        assert r["log"] == [("segment", 2), ("segment", 1), ("backward_end", -1), ("segment", 0)], r["log"]
        assert r["same_grad"] and r["mean_ok"], "averaged gradient must equal the mean of the ranks' local gradients on every rank"
        assert r["covered"] > 0.99 and r["upd_ok"] and r["n_losses"] == 6
        # staged backward (the captured step's form): stage k, then segment k's collective, then stage k+1 - and every segment's launch sees
        # only ITS OWN and the later segments' gradients present (the earlier layers' backward has not been issued yet)
        sg = r["staged"]
        assert sg["log"] == [("stage", "head"), ("segment", 2), ("stage", "encoder"), ("segment", 1), ("stage", "backbone"), ("segment", 0)], sg["log"]
        assert [i for i, _ in sg["events"]] == [2, 1, 0] and sg["events"][0][1] == (False, False, True) and sg["events"][1][1] == (False, True, True), sg["events"]
        assert sg["red_log"] == [("segment", 2), ("segment", 1), ("segment", 0), ("backward_end", -1)], sg["red_log"]
        assert sg["equal"] and sg["p_equal"], "the staged backward must produce the hooked backward's averaged gradient and update, bit for bit"
        # num_boxes = all-reduced target count / world size (fai_detr/modelling.py:568-570): ranks hold 4 and 8 targets -> 6
        assert r["num_boxes"] == [6.0, 6.0], r["num_boxes"]
