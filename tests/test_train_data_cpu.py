"""Data-parallel partitioning contract and EMA on the CPU: the mirrors in focoos_amd/train_data.py vs the real
reference classes (imported from /root/reference when present) and their defining properties."""
import itertools
import math

import pytest
import torch

from focoos_amd.train_data import FlatEMA, InferenceSampler, TrainingSampler, lr_factor, per_rank_batch_size, rank_seed


def take(it, n):
    return list(itertools.islice(iter(it), n))


def test_training_sampler_ranks_partition_one_global_stream():
    size, world = 37, 4
    streams = [take(TrainingSampler(size, True, seed=5, rank=r, world_size=world), 50) for r in range(world)]
    merged = [streams[i % world][i // world] for i in range(200)]
    single = take(TrainingSampler(size, True, seed=5, rank=0, world_size=1), 200)
    assert merged == single                                   # ranks stride ONE identically seeded stream
    for e in range(5):                                         # which is a sequence of permutations of range(size)
        assert sorted(single[e * size:(e + 1) * size]) == list(range(size))
    assert take(TrainingSampler(size, False, rank=1, world_size=2), 4) == [1, 3, 5, 7]
    with pytest.raises(ValueError):
        TrainingSampler(0)
    with pytest.raises(TypeError):
        TrainingSampler(3.0)


def test_inference_sampler_and_batch_split():
    for size, world in ((10, 4), (8, 8), (3, 5), (1000, 7)):
        parts = [list(InferenceSampler(size, rank=r, world_size=world)) for r in range(world)]
        assert sum(parts, []) == list(range(size))
        assert max(map(len, parts)) - min(map(len, parts)) <= 1
    assert per_rank_batch_size(128, 8) == 16
    with pytest.raises(ValueError):
        per_rank_batch_size(100, 8)
    assert rank_seed(42, 3) == 45


def test_samplers_match_reference_when_available():
    from oracle import ref_import

    if not ref_import.reference_available():
        pytest.skip("reference tree not mounted")
    ref_import.install()
    import focoos.data.samplers as S
    from focoos.utils.distributed import comm

    orig = (comm.get_rank, comm.get_world_size)
    try:
        for rank, world in ((0, 1), (2, 4), (7, 8)):
            comm.get_rank, comm.get_world_size = (lambda r=rank: r), (lambda w=world: w)
            assert take(S.TrainingSampler(101, True, seed=9), 300) == take(TrainingSampler(101, True, seed=9, rank=rank, world_size=world), 300)
            assert list(S.InferenceSampler(101)) == list(InferenceSampler(101, rank=rank, world_size=world))
    finally:
        comm.get_rank, comm.get_world_size = orig


def test_flat_ema_matches_reference_updater():
    g = torch.Generator().manual_seed(0)
    flat = torch.randn(1000, generator=g)
    views = {"a.weight": flat[:600].view(20, 30), "b.bias": flat[600:].view(400)}
    bufs = {"bn.running_mean": torch.randn(8, generator=g), "bn.num_batches_tracked": torch.tensor(3)}
    ema = FlatEMA(flat, views, bufs, decay=0.999, warmups=20)
    ref = {n: v.clone() for n, v in list(views.items()) + list(bufs.items())}
    for step in range(1, 6):
        flat.add_(torch.randn(1000, generator=g) * 0.1)          # "optimizer step" through the flat buffer: the views follow
        bufs["bn.running_mean"].add_(0.05)
        bufs["bn.num_batches_tracked"].add_(1)
        d = ema.update()
        assert abs(d - 0.999 * (1 - math.exp(-step / 20))) < 1e-12
        for n, v in list(views.items()) + list(bufs.items()):      # EMAUpdater.update arithmetic (ema.py:106-137)
            if v.dtype == torch.float32:
                ref[n] = ref[n] * d + v * (1 - d)
            else:
                ref[n] = (ref[n] * d + v * (1.0 - d)).to(v.dtype)
    sd = ema.state_dict()
    assert sorted(sd) == sorted(ref)
    for n in ref:
        torch.testing.assert_close(sd[n], ref[n], rtol=1e-5, atol=1e-6)
    from oracle import ref_import

    if ref_import.reference_available():                          # and against the real EMAUpdater on a small nn.Module
        ref_import.install()
        import importlib.util
        import os
        import sys
        import types

        # ema.py imports focoos.trainer.hooks (-> checkpointer -> iopath, not installed): load the file with HookBase stubbed
        saved = {k: sys.modules.get(k) for k in ("focoos.trainer", "focoos.trainer.hooks")}
        pkg = types.ModuleType("focoos.trainer")
        pkg.__path__ = []
        hooks = types.ModuleType("focoos.trainer.hooks")
        hooks.HookBase = object
        sys.modules["focoos.trainer"], sys.modules["focoos.trainer.hooks"] = pkg, hooks
        try:
            spec = importlib.util.spec_from_file_location("_ref_ema", os.path.join(ref_import.REFERENCE_ROOT, "focoos/trainer/solver/ema.py"))
            mod = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(mod)
        finally:
            for k, v in saved.items():
                if v is None:
                    sys.modules.pop(k, None)
                else:
                    sys.modules[k] = v
        EMAState, EMAUpdater = mod.EMAState, mod.EMAUpdater

        m = torch.nn.Sequential(torch.nn.Linear(4, 3), torch.nn.BatchNorm1d(3))
        names = [n for n, _ in m.named_parameters()]
        flat2 = torch.cat([p.detach().reshape(-1) for p in m.parameters()]).clone()
        off, views2 = 0, {}
        for n, p in m.named_parameters():
            views2[n] = flat2[off:off + p.numel()].view(p.shape)
            p.data = views2[n]
            off += p.numel()
        bufs2 = dict(m.named_buffers())
        up = EMAUpdater(EMAState(), decay=0.99, warmups=5)
        up.init_state(m)
        mine = FlatEMA(flat2, views2, bufs2, decay=0.99, warmups=5)
        for _ in range(4):
            flat2.add_(0.01)
            m(torch.randn(6, 4))                                  # train-mode BN moves the running statistics
            up.update(m)
            mine.update()
        sd2 = mine.state_dict()
        for n in names + list(bufs2):
            torch.testing.assert_close(sd2[n].float(), up.state.state[n].float(), rtol=1e-5, atol=1e-6)


def test_lr_factor_matches_reference_schedulers():
    """lr_factor vs the reference's LRScheduler classes stepped like the trainer does (one scheduler.step() per iteration)."""
    cases = [("POLY", dict(warmup_factor=0.001, warmup_iters=20, warmup_method="linear", power=0.9, constant_ending=0.0)),
             ("POLY", dict(warmup_factor=1.0, warmup_iters=0, power=2.0, constant_ending=0.05)),
             ("COSINE", dict(warmup_factor=0.01, warmup_iters=15, warmup_method="quadratic")),
             ("MULTISTEP", dict(milestones=[0.5, 0.8], gamma=0.1, warmup_factor=0.1, warmup_iters=10, warmup_method="constant")),
             ("FIXED", dict())]
    # defining properties, available everywhere
    assert lr_factor("POLY", 0, 100, warmup_factor=0.001, warmup_iters=20) == pytest.approx(0.001)
    assert lr_factor("COSINE", 100, 100) == pytest.approx(0.0, abs=1e-12) and lr_factor("FIXED", 7, 10) == 1.0
    assert lr_factor("MULTISTEP", 80, 100, milestones=[0.5, 0.8]) == pytest.approx(0.01)
    with pytest.raises(NotImplementedError):
        lr_factor("STEP", 0, 10)
    from oracle import ref_import

    if not ref_import.reference_available():
        pytest.skip("reference tree not mounted")
    ref_import.install()
    import importlib.util
    import os

    spec = importlib.util.spec_from_file_location("_ref_lrs", os.path.join(ref_import.REFERENCE_ROOT, "focoos/trainer/solver/lr_scheduler.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    classes = {"POLY": mod.WarmupPolyLR, "COSINE": mod.WarmupCosineLR, "MULTISTEP": mod.WarmupMultiStepLR, "FIXED": mod.BaseLRScheduler}
    max_iters = 120
    for name, extra in cases:
        p1, p2 = torch.nn.Parameter(torch.zeros(1)), torch.nn.Parameter(torch.zeros(1))
        opt = torch.optim.SGD([{"params": [p1], "lr": 1e-4}, {"params": [p2], "lr": 1e-5}], lr=1e-4)
        sch = classes[name](optimizer=opt, max_iters=max_iters, **extra)
        for it in range(max_iters):
            f = lr_factor(name, it, max_iters, **extra)
            got = [g["lr"] for g in opt.param_groups]
            assert got[0] == pytest.approx(1e-4 * f, rel=1e-12, abs=1e-18) and got[1] == pytest.approx(1e-5 * f, rel=1e-12, abs=1e-18), (name, it)
            opt.step()
            sch.step()


@pytest.mark.parametrize("model_name,family", [("fai-detr-l-coco", "fai_detr"), ("fai-mf-l-coco-ins", "fai_mf"), ("bisenetformer-l-ade", "bisenetformer")])
def test_optimizer_hyperparams_match_reference_param_groups(model_name, family):
    """Per-parameter (lr, weight_decay) vs the reference's get_optimizer_params on the reference model itself - in particular
    that only parameters of normalisation MODULES lose their weight decay (biases of Linear / conv layers keep it)."""
    from oracle import ref_import

    if not ref_import.reference_available():
        pytest.skip("reference tree not mounted")
    ref_import.install()
    import importlib.util
    import os
    import sys
    import types

    from focoos_amd.registry import ModelRegistry
    from focoos_amd.state_spec import state_spec
    from focoos_amd.train_data import optimizer_hyperparams

    cfg = ModelRegistry.get_model_info(model_name)["config"]
    build = {"fai_detr": ref_import.build_reference_detr, "fai_mf": ref_import.build_reference_mf, "bisenetformer": ref_import.build_reference_bf}[family]
    rc = {k: v for k, v in cfg.items() if k != "resolution"} if family == "bisenetformer" else cfg
    model = build(rc)[0]
    # solver/build.py imports lr_scheduler (fine) and the ConvNext layer norm; load it standalone
    src = open(os.path.join(ref_import.REFERENCE_ROOT, "focoos/trainer/solver/build.py")).read()
    mod = types.ModuleType("_ref_solver_build")
    mod.__package__ = "focoos.trainer.solver"
    sys.modules.setdefault("focoos.trainer", types.ModuleType("focoos.trainer")).__path__ = []
    pkg = sys.modules.setdefault("focoos.trainer.solver", types.ModuleType("focoos.trainer.solver"))
    pkg.__path__ = [os.path.join(ref_import.REFERENCE_ROOT, "focoos/trainer/solver")]
    try:
        exec(compile(src, "build.py", "exec"), mod.__dict__)
    finally:
        for k in ("focoos.trainer", "focoos.trainer.solver", "focoos.trainer.solver.lr_scheduler"):
            sys.modules.pop(k, None)
    groups = mod.get_optimizer_params(model, base_lr=1e-4, weight_decay=0.02, weight_decay_norm=0.0, weight_decay_embed=0.005,
                                      backbone_multiplier=0.1, decoder_multiplier=0.5, head_multiplier=2.0)
    by_id = {id(g["params"][0]): (g["lr"], g["weight_decay"]) for g in groups}
    spec = state_spec(cfg, family)
    checked = 0
    for name, p in model.named_parameters():
        if id(p) not in by_id:
            continue
        lr, wd = optimizer_hyperparams(name, spec[name][1], 1e-4, 0.02, 0.0, 0.005, 0.1, 0.5, 2.0)
        assert (lr, wd) == pytest.approx(by_id[id(p)], rel=1e-12), (name, spec[name][1], (lr, wd), by_id[id(p)])
        checked += 1
    assert checked == len(groups) and checked > 300
