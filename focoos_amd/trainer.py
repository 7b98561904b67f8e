"""The engine's side of ``FocoosModel.train`` (focoos/models/focoos_model.py:221-274 -> focoos/trainer/trainer.py ``run_train`` :60-130,
``TrainerLoop.run_step`` :723-773): one process per GPU, every rank steps ``TrainStep`` (HIP autograd graph + fused AdamW, gradients
averaged with one bucketed RCCL all-reduce) over its shard of the total batch, rank 0 writes the artifacts the reference writes
(``model_final.pth`` with the reference's state-dict keys, ``model_info.json``).

What is mirrored: the data-parallel partitioning (``TrainerArgs.batch_size`` is the TOTAL batch; ``TrainingSampler`` strides one
identically seeded permutation by rank; rank RNG seed = seed + rank), the optimizer / schedule / EMA hyper-parameters, freeze_bn vs
batch statistics (SyncBN when num_gpus > 1, as ``trainer.py:333-334`` converts), the artifact names.  What is not: hub syncing,
visualisation hooks, periodic COCO evaluation, early stopping (trainer-side features outside SURVEY §8)."""
from __future__ import annotations

import json
import os
import time
import warnings
from collections import OrderedDict
from dataclasses import asdict
from typing import Dict, List, Optional

import torch

from .ports import TrainerArgs

WEIGHTS_NAME = "model_final.pth"    # focoos.ports.ArtifactName.WEIGHTS
INFO_NAME = "model_info.json"       # focoos.ports.ArtifactName.INFO


def _dist():
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


class ModelSnapshot:
    """What ``run_train`` reads from the engine-backed model, in picklable form.  ``launch`` starts its ranks with ``spawn``, which
    pickles the arguments: the engine (ctypes library handle, HIP streams, captured graphs, device tensors) cannot travel, so the
    ranks receive the family, the config and the fp32 state dict on the CPU and build their own training graph from them."""

    def __init__(self, model):
        self.family = getattr(model, "family", "fai_detr")
        self.config = dict(model.config)
        self._state = OrderedDict((k, v.detach().cpu()) for k, v in model.state_dict().items())
        self.device = torch.device("cuda", 0)

    def state_dict(self):
        return OrderedDict(self._state)


def check_supported(args: TrainerArgs) -> None:
    """TrainerArgs fields that change the training arithmetic of the reference and that this trainer does not implement are refused,
    never ignored (fields that only steer hub syncing / visualisation / periodic evaluation / early stopping are accepted: see the
    module docstring).  Reference: ``build_optimizer`` solver/build.py:39-138, ``TrainerLoop`` trainer/trainer.py:600-773."""
    if str(args.optimizer).upper() != "ADAMW":
        raise NotImplementedError(f"TrainerArgs.optimizer={args.optimizer!r}: the engine's optimizer is the fused AdamW (fx_adamw_step_f32)")
    if args.optimizer_extra:
        raise NotImplementedError("TrainerArgs.optimizer_extra is not supported by the fused AdamW step")
    if args.resume:
        raise NotImplementedError("TrainerArgs.resume: periodic checkpoints are not written, so there is nothing to resume from "
                                  "(use init_checkpoint with a model_final.pth)")
    if float(args.decoder_multiplier) != 1.0 or float(args.head_multiplier) != 1.0:
        raise NotImplementedError("TrainerArgs.decoder_multiplier / head_multiplier != 1.0 are not supported (only backbone_multiplier)")
    if not args.amp_enabled:
        warnings.warn("TrainerArgs.amp_enabled=False: the engine always computes in bf16 with fp32 master weights")


def _spawned_rank(args, data_train, data_val, snapshot, family, config, im_size, model_info, hub):
    """Body of one ``launch`` worker: the processor is rebuilt from its configuration (it may hold device buffers)."""
    from .model import ProcessorManager

    processor = ProcessorManager.get_processor(family, config, image_size=im_size)
    return run_train(args, data_train, data_val, snapshot, processor, model_info, hub)


def run_train(args: TrainerArgs, data_train, data_val, model, processor, model_info, hub=None) -> Dict[str, float]:
    """Per-rank training body.  ``model`` is the engine-backed FAIDetr / BisenetFormer mirror or its ``ModelSnapshot`` (its state_dict
    seeds the trainable graph); ``data_train[i]`` is a DatasetEntry.  Returns the last step's losses (floats) on every rank.
    Ignored TrainerArgs fields (trainer-side features outside SURVEY §8): ckpt_dir, checkpointer_period / _max_to_keep, eval_period,
    samples, early_stop, patience, workers, ddp_*, gather_metric_period, zero_grad_before_forward, sync_to_hub, size_divisibility."""
    check_supported(args)
    from .train_data import TrainingSampler, per_rank_batch_size, rank_seed
    from .train_detr import FAIDetrTrainable, TrainStep

    family = getattr(model, "family", "fai_detr")
    if family == "fai_detr":
        trainable = FAIDetrTrainable
    elif family == "bisenetformer":
        from .train_bf import BisenetFormerTrainable as trainable
    elif family == "fai_mf":
        from .train_mf import FAIMaskFormerTrainable as trainable
    else:
        raise NotImplementedError(f"no training graph for model family {family!r}")
    rank, world = _dist()
    local = int(os.environ.get("LOCAL_RANK", rank))
    dev = torch.device("cuda", local if world > 1 else (model.device.index or 0))
    torch.cuda.set_device(dev)
    torch.manual_seed(rank_seed(args.seed, rank))
    norm = "FrozenBN" if args.freeze_bn else ("SyncBN" if world > 1 else "BN")
    # 16-bit element type of the step: bf16 by default; FX_TRAIN_DTYPE=fp16 with amp_enabled = the reference's own mixed precision (fp16
    # autocast + GradScaler(init_scale=2**10), trainer/trainer.py:645,735-773): fp16 library + dynamic loss scale inside the optimizer launch
    from . import _lib

    dtype = os.environ.get("FX_TRAIN_DTYPE", "bf16") if args.amp_enabled else "bf16"
    prev_dtype = _lib.set_compute_dtype(dtype)
    try:     # the run's element type is restored whatever ends the run (a raising stepper.check() used to leave the process in fp16)
        net = trainable(model.config, norm=norm).to(dev)
        net.load_state_dict(model.state_dict(), strict=True)
        if args.init_checkpoint:
            state = torch.load(args.init_checkpoint, map_location="cpu", weights_only=True)
            net.load_state_dict(state.get("model", state), strict=False)
        net.train()
        stepper = TrainStep(net, lr=args.learning_rate, backbone_multiplier=args.backbone_multiplier, weight_decay=args.weight_decay,
                            weight_decay_norm=args.weight_decay_norm, weight_decay_embed=args.weight_decay_embed, max_grad_norm=args.clip_gradients,
                            ema_decay=args.ema_decay if args.ema_enabled else None, ema_warmups=args.ema_warmup, scheduler=args.scheduler,
                            max_iters=args.max_iters, scheduler_extra=args.scheduler_extra)
        bs = per_rank_batch_size(args.batch_size, world)
        sampler = iter(TrainingSampler(len(data_train), shuffle=True, seed=args.seed, rank=rank, world_size=world))
        processor.train(True)
        losses: Dict[str, torch.Tensor] = {}
        t0 = time.perf_counter()
        for it in range(args.max_iters):
            entries = [data_train[next(sampler)] for _ in range(bs)]
            images, targets = processor.preprocess(entries, device=dev)
            losses = stepper.step(images, targets)
            if args.log_period and (it + 1) % args.log_period == 0:
                stepper.check()    # every rank (MAX-all-reduced status): a diverged step (NaN / inf matching costs) ends the run as SciPy's error does in the reference
            if rank == 0 and args.log_period and (it + 1) % args.log_period == 0:
                tot = float(sum(v.detach().float() for v in losses.values()))
                print(f"[focoos_amd.train] iter {it + 1}/{args.max_iters} total_loss {tot:.4f} {(time.perf_counter() - t0) / (it + 1) * 1e3:.1f} ms/iter", flush=True)
        torch.cuda.synchronize(dev)
        stepper.check()
    finally:
        _lib.set_compute_dtype(prev_dtype)
    out = {k: float(v.detach().float()) for k, v in losses.items()}
    if rank == 0:
        folder = os.path.join(args.output_dir, args.run_name.strip())
        os.makedirs(folder, exist_ok=True)
        state = stepper.ema.state_dict() if (args.ema_enabled and stepper.ema is not None) else {k: v.detach().cpu() for k, v in net.state_dict().items()}
        net_state = {k: v.detach().cpu() for k, v in net.state_dict().items()}
        net_state.update({k: v.detach().cpu() for k, v in state.items() if k in net_state})
        order = list(model.state_dict())            # the reference's key order (state_spec), as torch.save of the reference module writes it
        full = {k: net_state[k] for k in order if k in net_state}
        full.update({k: v for k, v in net_state.items() if k not in full})
        torch.save({"model": full, "iteration": args.max_iters}, os.path.join(folder, WEIGHTS_NAME))
        info = {k: getattr(model_info, k) for k in ("name", "model_family", "classes", "im_size", "task", "config", "description")}
        info.update(name=args.run_name.strip(), weights_uri=os.path.join(folder, WEIGHTS_NAME), status="TRAINING_COMPLETED",
                    train_args={k: v for k, v in asdict(args).items()}, final_losses=out)
        with open(os.path.join(folder, INFO_NAME), "w") as f:
            json.dump(info, f, indent=1, default=str)
    if world > 1:
        import torch.distributed as dist

        dist.barrier()
    processor.eval()
    return out


def train(focoos_model, args: TrainerArgs, data_train, data_val=None, hub=None):
    """FocoosModel.train (focoos_model.py:221-274): checks, one process per GPU through ``launch`` when num_gpus > 1, then reloads the
    trained weights into the inference engine and returns to eval mode."""
    from .launch import launch

    assert len(data_train) > 0, "empty training set"
    assert args.num_gpus, "Training without GPUs is not supported. num_gpus must be greater than 0"
    check_supported(args)
    if args.num_gpus > 1:
        launch(_spawned_rank, args.num_gpus, dist_url="auto",
               args=(args, data_train, data_val, ModelSnapshot(focoos_model.model), focoos_model.model.family, dict(focoos_model.model_info.config),
                     focoos_model.model_info.im_size, focoos_model.model_info, hub))
    else:
        run_train(args, data_train, data_val, focoos_model.model, focoos_model.processor, focoos_model.model_info, hub)
    folder = os.path.join(args.output_dir, args.run_name.strip())
    model_path, info_path = os.path.join(folder, WEIGHTS_NAME), os.path.join(folder, INFO_NAME)
    if not os.path.exists(model_path):
        raise FileNotFoundError(f"Training did not end correctly, model file not found at {model_path}")
    if not os.path.exists(info_path):
        raise FileNotFoundError(f"Training did not end correctly, metadata file not found at {info_path}")
    state = torch.load(model_path, map_location="cpu", weights_only=True)
    focoos_model.model.load_state_dict(state["model"], strict=False)
    focoos_model.model_info.name = args.run_name.strip()
    focoos_model.model_info.weights_uri = model_path
    focoos_model.model.eval()
    focoos_model.processor.eval()
    return focoos_model
