"""Host-side pieces of the data-parallel training contract that sit around TrainStep: how a global batch is
partitioned over ranks and the weight EMA of the reference's trainer.  Pure host / torch logic (no kernels), mirrored so that a
run on N GPUs sees the same sample order and the same EMA trajectory as the reference:

  TrainingSampler / InferenceSampler    focoos/data/samplers.py:10-65, 68-104   (identically seeded permutation, strided by rank)
  per_rank_batch_size                   focoos/data/loaders.py:61-65            (TrainerArgs.batch_size is the TOTAL batch)
  rank_seed                             focoos/trainer/trainer.py:101           (seed + rank)
  FlatEMA                               focoos/trainer/solver/ema.py:83-137     (EMAUpdater: decay * (1 - exp(-updates / warmups)))
  lr_factor                             focoos/trainer/solver/lr_scheduler.py:19-171 (FIXED / POLY / COSINE / MULTISTEP with warm-up)
"""
from __future__ import annotations

import itertools
import math
from typing import Dict, Iterator, Optional

import torch


def _rank_world(rank: Optional[int], world_size: Optional[int]):
    if rank is not None and world_size is not None:
        return int(rank), int(world_size)
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def per_rank_batch_size(total_batch_size: int, world_size: int) -> int:
    if total_batch_size <= 0 or total_batch_size % world_size != 0:
        raise ValueError(f"Total batch size ({total_batch_size}) must be divisible by the number of gpus ({world_size}).")
    return total_batch_size // world_size


def rank_seed(seed: int, rank: int) -> int:
    return int(seed) + int(rank)


class TrainingSampler:
    """Infinite stream ``shuffle(range(size)) + shuffle(range(size)) + ...`` from ONE generator seeded identically on every rank;
    rank r takes elements r, r + world, r + 2*world, ... of that stream - the ranks' batches are disjoint slices of one global order."""

    def __init__(self, size: int, shuffle: bool = True, seed: int = 0, rank: Optional[int] = None, world_size: Optional[int] = None):
        if not isinstance(size, int):
            raise TypeError(f"TrainingSampler(size=) expects an int. Got type {type(size)}.")
        if size <= 0:
            raise ValueError(f"TrainingSampler(size=) expects a positive int. Got {size}.")
        self._size, self._shuffle, self._seed = size, shuffle, int(seed)
        self._rank, self._world_size = _rank_world(rank, world_size)

    def __iter__(self) -> Iterator[int]:
        yield from itertools.islice(self._infinite_indices(), self._rank, None, self._world_size)

    def _infinite_indices(self):
        g = torch.Generator()
        g.manual_seed(self._seed)
        while True:
            if self._shuffle:
                yield from torch.randperm(self._size, generator=g).tolist()
            else:
                yield from torch.arange(self._size).tolist()


class InferenceSampler:
    """Contiguous shards of range(size), the first ``size % world`` ranks one sample longer: every sample exactly once."""

    def __init__(self, size: int, rank: Optional[int] = None, world_size: Optional[int] = None):
        assert size > 0
        self._size = size
        self._rank, self._world_size = _rank_world(rank, world_size)
        self._local_indices = self._get_local_indices(size, self._world_size, self._rank)

    @staticmethod
    def _get_local_indices(total_size: int, world_size: int, rank: int) -> range:
        shard, left = divmod(total_size, world_size)
        sizes = [shard + int(r < left) for r in range(world_size)]
        begin = sum(sizes[:rank])
        return range(begin, min(begin + sizes[rank], total_size))

    def __iter__(self):
        yield from self._local_indices

    def __len__(self):
        return len(self._local_indices)


class FlatEMA:
    """EMAUpdater over the optimizer's ONE flat fp32 parameter buffer: ``ema = decay_t * ema + (1 - decay_t) * p`` as a single
    elementwise pass (the reference issues a _foreach_mul_ / _foreach_add_ pair over ~500 tensors) with the reference's warm-up
    ``decay_t = decay * (1 - exp(-updates / warmups))``.  ``buffers`` (BatchNorm running statistics, live-BN mode) are averaged
    the same way, integer buffers (num_batches_tracked) follow the reference's ``ema * d + val * (1 - d)`` then cast back."""

    def __init__(self, flat_params: torch.Tensor, named_views: Dict[str, torch.Tensor], buffers: Optional[Dict[str, torch.Tensor]] = None,
                 decay: float = 0.999, warmups: int = 2000):
        self.decay, self.warmups, self.updates = decay, warmups, 0
        self.flat_p = flat_params
        self.flat = flat_params.detach().clone()
        base = flat_params.data_ptr()
        self.views = {}
        for n, v in named_views.items():   # EMA views at the same offsets as the parameter views
            off = (v.data_ptr() - base) // flat_params.element_size()
            self.views[n] = self.flat[off:off + v.numel()].view(v.shape)
        self.buffers = buffers or {}
        self.buf_state = {n: b.detach().clone() for n, b in self.buffers.items()}

    def decay_at(self, updates: int) -> float:
        return self.decay * (1 - math.exp(-updates / self.warmups)) if self.warmups > 0 else self.decay

    @torch.no_grad()
    def update(self) -> float:
        self.updates += 1
        d = self.decay_at(self.updates)
        self.flat.mul_(d).add_(self.flat_p, alpha=1 - d)
        for n, b in self.buffers.items():
            e = self.buf_state[n]
            if b.dtype in (torch.float32, torch.float16):
                e.mul_(d).add_(b, alpha=1 - d)
            else:
                e.copy_(e * d + b * (1.0 - d))
        return d

    def state_dict(self) -> Dict[str, torch.Tensor]:
        """name -> tensor like the reference's EMAState.state (parameters and buffers)."""
        out = {n: v.clone() for n, v in self.views.items()}
        out.update({n: v.clone() for n, v in self.buf_state.items()})
        return out


# ------------------------------------------------------------------------------------------------ learning-rate schedules
def _warmup_factor(method: str, it: int, warmup_iters: int, warmup_factor: float) -> float:
    """focoos/trainer/solver/lr_scheduler.py:142-171."""
    if it >= warmup_iters:
        return 1.0
    if method == "constant":
        return warmup_factor
    if method == "linear":
        alpha = it / warmup_iters
        return warmup_factor * (1 - alpha) + alpha
    if method == "quadratic":
        alpha = (it / warmup_iters) ** 2
        return warmup_factor * (1 - alpha) + alpha
    raise ValueError("Unknown warmup method: {}".format(method))


def lr_factor(name: str, it: int, max_iters: int, *, milestones=(), gamma: float = 0.1, warmup_factor: float = 1.0, warmup_iters: int = 0,
              warmup_method: str = "linear", power: float = 0.9, constant_ending: float = 0.0) -> float:
    """Multiplier of every parameter group's base learning rate at iteration ``it`` - WarmupPolyLR / WarmupMultiStepLR /
    WarmupCosineLR / the constant base scheduler of focoos/trainer/solver/lr_scheduler.py:19-139 (selected by name like
    build_lr_scheduler, solver/build.py:141-159).  The engine applies it with one device-side multiply of the optimizer's
    per-chunk learning-rate table (FlatAdamW.set_lr_scale) instead of a scheduler object stepping ~500 param groups."""
    from bisect import bisect_right

    key = name.upper()
    w = _warmup_factor(warmup_method, it, warmup_iters, warmup_factor)
    if key == "POLY":
        decay = math.pow((1.0 - it / max_iters), power)
        if constant_ending > 0 and w == 1.0 and decay < constant_ending:
            return constant_ending
        return w * decay
    if key == "MULTISTEP":
        if list(milestones) != sorted(milestones):
            raise ValueError("Milestones should be a list of increasing integers. Got {}", milestones)
        ms = [int(m * max_iters) for m in milestones]
        return w * gamma ** bisect_right(ms, it)
    if key == "COSINE":
        return w * 0.5 * (1.0 + math.cos(math.pi * it / max_iters))
    if key == "FIXED":
        return 1.0
    raise NotImplementedError(f"Scheduler {name} is not supported.")


# ------------------------------------------------------------------------------------------------ optimizer hyper-parameters
_NORM_KINDS = ("bn_w", "bn_b", "ln_w", "ln_b")


def optimizer_hyperparams(name: str, kind: str, base_lr: float, weight_decay: float, weight_decay_norm: float = 0.0,
                          weight_decay_embed: float = 0.0, backbone_multiplier: float = 0.1, decoder_multiplier: float = 1.0,
                          head_multiplier: float = 1.0):
    """(lr, weight_decay) of one parameter exactly as get_optimizer_params assigns them (focoos/trainer/solver/build.py:39-101):
    lr x backbone_multiplier for modules under "backbone", x decoder_multiplier under "pixel_decoder" (the backbone lives INSIDE
    pixel_decoder, so it gets both), x head_multiplier under "head" unless the module name contains "classifier"; weight decay
    = weight_decay_norm for parameters of normalisation MODULES (BatchNorm / LayerNorm - not for every 1-D tensor: Linear and conv
    biases keep the full decay), weight_decay_embed for nn.Embedding weights.  ``kind`` is the parameter's kind in
    focoos_amd.state_spec (bn_w, ln_b, emb, lin_b, ...), which encodes the owning module's type."""
    module_name = name.rsplit(".", 1)[0] if "." in name else ""
    lr, wd = base_lr, weight_decay
    if "backbone" in module_name:
        lr *= backbone_multiplier
        if backbone_multiplier == 0:
            wd = 0.0
    if "pixel_decoder" in module_name:
        lr *= decoder_multiplier
        if backbone_multiplier == 0:   # (sic) the reference tests backbone_multiplier here as well
            wd = 0.0
    if "head" in module_name and "classifier" not in module_name:
        lr *= head_multiplier
        if head_multiplier == 0:
            wd = 0.0
    if kind in _NORM_KINDS:
        wd = weight_decay_norm
    if kind == "emb" or "pos_embed" in name.rsplit(".", 1)[-1]:
        wd = weight_decay_embed
    return lr, wd
