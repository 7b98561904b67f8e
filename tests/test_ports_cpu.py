"""The container contract of the boundary types (``forward -> ModelOutput`` "dataclass-dict"): the cases of the reference's
tests/test_ports.py (DictClass: field / key / index access, to_tuple, the two assignment paths, mapping behaviour, __reduce__, error
cases) run against focoos_amd.ports.DictClass, the three model-output types and DatasetEntry - and, where /root/reference is present, the
same operations on the reference's own class side by side."""
import pickle
from dataclasses import dataclass
from typing import Optional

import pytest
import torch

from focoos_amd.ports import BisenetFormerOutput, DatasetEntry, DETRModelOutput, DictClass, MaskFormerModelOutput, ModelOutput


@dataclass
class Sample(DictClass):
    name: str
    value: int
    optional_field: Optional[str] = None
    default_field: str = "default"


def test_fields_keys_and_indices_are_one_view():
    o = Sample(name="a", value=42)
    assert (o.name, o.value, o.optional_field, o.default_field) == ("a", 42, None, "default")
    assert (o["name"], o["value"], o["optional_field"], o["default_field"]) == ("a", 42, None, "default")
    assert len(o) == 4 and list(o.keys()) == ["name", "value", "optional_field", "default_field"]
    assert ("name", "a") in o.items() and 42 in o.values() and "optional_field" in o
    t = o.to_tuple()
    assert t == ("a", 42, "default") and o[0] == "a" and o[1] == 42 and o[-1] == "default" and o[:2] == ("a", 42)   # None fields dropped
    assert Sample(name="b", value=0, optional_field="x").to_tuple() == ("b", 0, "x", "default")


def test_both_assignment_paths_update_both_views():
    o = Sample(name="a", value=1)
    o.name = "n2"
    o.optional_field = "opt"
    assert o["name"] == "n2" and o["optional_field"] == "opt" and o.to_tuple() == ("n2", 1, "opt", "default")
    o["value"] = 9
    o["name"] = "n3"
    assert o.value == 9 and o.name == "n3"
    o.optional_field = None          # the reference leaves the mapping entry as it was when None is assigned by attribute
    assert o.optional_field is None and o["optional_field"] == "opt"
    o.optional_field = "back"
    assert o["optional_field"] == "back"


def test_errors_and_reduce():
    o = Sample(name="a", value=1)
    with pytest.raises(KeyError):
        o["nope"]
    with pytest.raises(IndexError):
        o[10]
    ctor, args, state = o.__reduce__()
    assert ctor == Sample.__new__ and args == (Sample,) and state == {"name": "a", "value": 1, "optional_field": None, "default_field": "default"}
    p = pickle.loads(pickle.dumps(o))
    assert isinstance(p, Sample) and p.name == "a" and p["value"] == 1 and list(p.keys()) == list(o.keys())

    @dataclass
    class Empty(DictClass):
        pass

    with pytest.raises(ValueError):
        Empty()


@pytest.mark.parametrize("cls,first", [(DETRModelOutput, "boxes"), (MaskFormerModelOutput, "masks"), (BisenetFormerOutput, "masks")])
def test_model_outputs_are_dataclass_dicts(cls, first):
    a, b = torch.zeros(2, 3, 4), torch.ones(2, 3, 5)
    out = cls(**{first: a, "logits": b, "loss": None})
    assert isinstance(out, ModelOutput) and isinstance(out, dict)
    assert list(out.keys()) == ["loss", first, "logits"]          # `loss` is ModelOutput's field: first, like in the reference
    assert out[first] is a and out["logits"] is b and out.loss is None and getattr(out, first) is a
    assert len(out.to_tuple()) == 2 and out[0] is a and out[1] is b      # inference: (boxes | masks, logits) - what an export traces
    out.loss = {"loss_vfl": torch.tensor(1.0)}
    assert out["loss"] is out.loss and len(out.to_tuple()) == 3 and out[0] is out.loss
    p = pickle.loads(pickle.dumps(cls(**{first: a, "logits": b, "loss": None})))
    assert torch.equal(p[first], a) and torch.equal(p.logits, b) and p["loss"] is None


def test_dataset_entry_is_a_mapping_too():
    e = DatasetEntry(image=torch.zeros(3, 8, 8), height=8, width=6)
    assert e["height"] == 8 and e.get("width") == 6 and e.get("missing") is None and e.instances is None
    assert e.to_tuple()[1:] == (8, 6)


def test_same_behaviour_as_the_reference_class():
    from oracle import ref_import

    if not ref_import.reference_available():
        pytest.skip("/root/reference not present")
    ref_import.install()
    from focoos.ports import DictClass as RefDictClass

    @dataclass
    class RefSample(RefDictClass):
        name: str
        value: int
        optional_field: Optional[str] = None
        default_field: str = "default"

    def script(o):
        log = [tuple(o.keys()), o.to_tuple(), o[0], o["value"], len(o)]
        o.name = "x"
        o["value"] = 5
        o.optional_field = "opt"
        log += [o.to_tuple(), o["name"], o.value, o[2]]
        o.optional_field = None
        log += [o.optional_field, o["optional_field"], o.to_tuple(), tuple(o.items())]
        for bad in ("nope", 10):
            try:
                o[bad]
                log.append("no error")
            except Exception as e:
                log.append(type(e).__name__)
        r = o.__reduce__()
        log += [r[1][0].__name__.replace("Ref", ""), r[2]]
        return log

    assert script(Sample(name="a", value=1)) == script(RefSample(name="a", value=1))
