"""The spawn boundary of FocoosModel.train(num_gpus > 1) on CPU: `launch()` starts its ranks with the `spawn` start method,
which pickles the arguments - the engine-backed model (ctypes handle, HIP streams, graphs) cannot travel, trainer.ModelSnapshot can.  Two
gloo ranks receive the snapshot through the real `launch()` and check what `run_train` reads from it."""
import pickle
from collections import OrderedDict

import pytest
import torch

from focoos_amd.launch import launch
from focoos_amd.ports import TrainerArgs
from focoos_amd.trainer import ModelSnapshot, check_supported


class _FakeEngineModel:
    """Stands in for model._EngineModel on a box without a GPU: same surface run_train reads, plus an unpicklable member like the engine."""
    family = "fai_detr"

    def __init__(self):
        self.config = {"num_classes": 3, "backbone_config": {"depth": 50}}
        self._sd = OrderedDict(a=torch.arange(6.0).reshape(2, 3), b=torch.ones(4))
        self.engine = lambda: None   # lambdas do not pickle - like the CDLL inside the real engine

    def state_dict(self):
        return OrderedDict((k, v.clone()) for k, v in self._sd.items())


def _rank_body(snapshot, expect_keys):
    import torch.distributed as dist

    assert dist.get_world_size() == 2
    sd = snapshot.state_dict()
    assert list(sd) == expect_keys and snapshot.family == "fai_detr" and snapshot.config["num_classes"] == 3
    t = sd["a"].sum().reshape(1)
    dist.all_reduce(t)
    assert float(t) == 2 * 15.0


def test_snapshot_pickles_and_engine_model_does_not():
    m = _FakeEngineModel()
    with pytest.raises(Exception):
        pickle.dumps(m)
    snap = pickle.loads(pickle.dumps(ModelSnapshot(m)))
    assert torch.equal(snap.state_dict()["a"], m.state_dict()["a"]) and snap.family == "fai_detr"


def test_two_ranks_receive_the_snapshot_through_launch():
    launch(_rank_body, 2, dist_url="auto", args=(ModelSnapshot(_FakeEngineModel()), ["a", "b"]), backend="gloo")


def test_unsupported_trainer_args_are_refused():
    check_supported(TrainerArgs(run_name="x"))
    for kw in (dict(optimizer="SGD"), dict(resume=True), dict(decoder_multiplier=0.5), dict(optimizer_extra={"momentum": 0.9})):
        with pytest.raises(NotImplementedError):
            check_supported(TrainerArgs(run_name="x", **kw))
