"""Two concurrent batch parts (the default since round 2, engine._MultiPlan) must be bit-identical to the SAME two parts run one after
the other on one stream.  Until round 2 they were not: kernels containing packed-fp32 VALU instructions compute wrong values in lanes
48-63 when waves of a second hardware queue share their CU (bisection scripts/dev/two_part_bisect.py, profiles/
r02_two_queue_*.txt).  The library is now compiled without that instruction class (focoos_amd/build.py); `FX_PK_F32=1 python -m
focoos_amd.build --force` rebuilds the reproducer (56-58 of 60 replays wrong)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_two_concurrent_batch_parts_equal_serial_parts():
    from focoos_amd.model import FAIDetr
    from focoos_amd.registry import ModelRegistry
    from focoos_amd.synth import synth_image_structured as sis

    cfg = ModelRegistry.get_model_info("fai-detr-l-obj365")["config"]
    eng = FAIDetr(cfg, device="cuda:0", seed=0).engine
    B = 32
    imgs = torch.from_numpy(np.stack([sis(100 + i) for i in range(B)])).to("cuda:0")
    pl = eng.plan(B, 640, 640, False, 2)
    st = eng.stream
    with torch.cuda.stream(st):
        pl.input.copy_(imgs)
        pl.sizes.copy_(torch.tensor([[640, 640]] * B, dtype=torch.int32))
        for p in pl.parts:
            p._launch(p.ops, st.cuda_stream, 0.3)
    st.synchronize()
    ref = (pl.probs.clone(), pl.boxes.clone(), pl.det_count.clone())
    bad = 0
    for _ in range(60):
        with torch.cuda.stream(st):
            pl.run(st.cuda_stream, 0.3, None, True)
        st.synchronize()
        bad += not (torch.equal(ref[0], pl.probs) and torch.equal(ref[1], pl.boxes) and torch.equal(ref[2], pl.det_count))
    assert bad == 0, f"{bad} of 60 concurrent replays differ from the serial result"


@pytest.mark.parametrize("family", ["fai_mf", "bisenetformer"])
def test_two_concurrent_batch_parts_equal_serial_parts_mask_families(family):
    """Same check for the MaskFormer (bs=8, 320x384) and BiSeNetFormer (bs=16, 384x512) engines: probabilities, low-resolution mask
    probabilities and packed detections of 40 concurrent replays equal the serial run of the same two parts."""
    from focoos_amd.model import BisenetFormer, FAIMaskFormer
    from focoos_amd.registry import ModelRegistry
    from focoos_amd.synth import synth_image_structured as sis

    if family == "fai_mf":
        cfg, cls, B, H, W = ModelRegistry.get_model_info("fai-mf-l-coco-ins")["config"], FAIMaskFormer, 8, 320, 384
    else:
        cfg, cls, B, H, W = ModelRegistry.get_model_info("bisenetformer-l-ade")["config"], BisenetFormer, 16, 384, 512
    eng = cls(cfg, device="cuda:0", seed=0).engine
    imgs = torch.from_numpy(np.stack([sis(200 + i, H, W) for i in range(B)])).to("cuda:0")
    pl = eng.plan(B, H, W, False, None, 2)
    assert getattr(pl, "n", 1) == 2
    st = eng.stream
    with torch.cuda.stream(st):
        pl.input.copy_(imgs)
        for p in pl.parts:
            p._launch(p.ops, st.cuda_stream, 0.3)
    st.synchronize()
    keys = ("probs", "mask_probs", "det_count", "det_scores", "det_boxes")
    ref = {k: getattr(pl, k).clone() for k in keys}
    bad = 0
    for _ in range(40):
        with torch.cuda.stream(st):
            pl.run(st.cuda_stream, 0.3, None, True)
        st.synchronize()
        bad += not all(torch.equal(ref[k], getattr(pl, k)) for k in keys)
    assert bad == 0, f"{bad} of 40 concurrent replays differ from the serial result"


def test_pipeline_lanes_equal_single_plan():
    """Throughput mode (engine.pipeline(), round 6): three DIFFERENT batches in flight on three whole-batch plans, 12 rounds of them
    back to back without a host synchronisation in between - every batch's probabilities, boxes and packed detections equal the
    one-batch-at-a-time result of the same images (a single whole-batch plan run alone), bit for bit; the lanes share nothing but the
    read-only weights, so concurrency must not change a value (the two-queue hazard of DESIGN 5 would show up exactly here)."""
    from focoos_amd.model import FAIDetr
    from focoos_amd.registry import ModelRegistry
    from focoos_amd.synth import synth_image_structured as sis

    cfg = ModelRegistry.get_model_info("fai-detr-l-obj365")["config"]
    eng = FAIDetr(cfg, device="cuda:0", seed=0).engine
    B, depth = 16, 3
    batches = [torch.from_numpy(np.stack([sis(300 + 40 * j + i) for i in range(B)])).to("cuda:0") for j in range(depth)]
    sizes = torch.tensor([[640, 640]] * B, dtype=torch.int32, device="cuda:0")
    keys = ("probs", "boxes", "det_count", "det_scores", "det_labels", "det_boxes")
    # reference: each batch alone on ONE plan, one stream, synchronised
    single = eng.plan(B, 640, 640, False, 1)
    st = eng.stream
    ref = []
    for x in batches:
        with torch.cuda.stream(st):
            single.input.copy_(x)
            single.sizes.copy_(sizes)
            single.run(st.cuda_stream, 0.3, None, True)
        st.synchronize()
        ref.append({k: getattr(single, k).clone() for k in keys})
    pipe = eng.pipeline(B, 640, 640, depth)
    assert pipe.depth == depth and len({s.cuda_stream for _, s in pipe.lanes}) == depth      # one stream per lane
    assert len({pl.input.data_ptr() for pl, _ in pipe.lanes}) == depth                           # own buffers per lane
    bad = 0
    for rnd in range(12):
        tickets = [pipe.submit(batches[(j + rnd) % depth], sizes, 0.3) for j in range(depth)]
        for j, t in enumerate(tickets):
            pl = pipe.wait(t)
            want = ref[(j + rnd) % depth]
            bad += not all(torch.equal(want[k], getattr(pl, k)) for k in keys)
    pipe.synchronize()
    assert bad == 0, f"{bad} of 36 pipelined batches differ from the one-batch-at-a-time result"
    assert int(ref[0]["det_count"].sum()) > 0 and not torch.equal(ref[0]["probs"], ref[1]["probs"])   # the comparison is not vacuous


@pytest.mark.parametrize("family", ["fai_mf", "bisenetformer"])
def test_pipeline_lanes_equal_single_plan_mask_families(family):
    """The throughput mode of the MaskFormer (bs=4, 320x384) and BiSeNetFormer (bs=8, 384x512) engines: three different batches in flight,
    8 rounds, every batch's class probabilities, low-resolution mask probabilities and packed detections equal the one-batch-at-a-time run."""
    from focoos_amd.model import BisenetFormer, FAIMaskFormer
    from focoos_amd.registry import ModelRegistry
    from focoos_amd.synth import synth_image_structured as sis

    if family == "fai_mf":
        cfg, cls, B, H, W = ModelRegistry.get_model_info("fai-mf-l-coco-ins")["config"], FAIMaskFormer, 4, 320, 384
    else:
        cfg, cls, B, H, W = ModelRegistry.get_model_info("bisenetformer-l-ade")["config"], BisenetFormer, 8, 384, 512
    eng = cls(cfg, device="cuda:0", seed=0).engine
    depth = 3
    batches = [torch.from_numpy(np.stack([sis(500 + 20 * j + i, H, W) for i in range(B)])).to("cuda:0") for j in range(depth)]
    sizes = torch.tensor([[H, W]] * B, dtype=torch.int32, device="cuda:0")
    keys = ("probs", "mask_probs", "det_count")
    packed = ("det_scores", "det_boxes")     # [B, Q, ...] with det_count[b] valid rows per image; the rows behind them are whatever an earlier batch left

    def same(want, pl):
        if not all(torch.equal(want[k], getattr(pl, k)) for k in keys):
            return False
        cnt = want["det_count"].tolist()
        return all(torch.equal(want[k][b, :n], getattr(pl, k)[b, :n]) for k in packed for b, n in enumerate(cnt))

    single = eng.plan(B, H, W, False, None, 1)
    st = eng.stream
    ref = []
    for x in batches:
        with torch.cuda.stream(st):
            single.input.copy_(x)
            single.sizes.copy_(sizes)
            single.run(st.cuda_stream, 0.3, None, True)
        st.synchronize()
        ref.append({k: getattr(single, k).clone() for k in keys + packed})
    pipe = eng.pipeline(B, H, W, depth)
    assert pipe.depth == depth and len({s.cuda_stream for _, s in pipe.lanes}) == depth
    bad = 0
    for rnd in range(8):
        tickets = [pipe.submit(batches[(j + rnd) % depth], sizes, 0.3) for j in range(depth)]
        for j, t in enumerate(tickets):
            pl = pipe.wait(t)
            bad += not same(ref[(j + rnd) % depth], pl)
    pipe.synchronize()
    assert bad == 0, f"{bad} of 24 pipelined batches differ from the one-batch-at-a-time result"
    assert not torch.equal(ref[0]["probs"], ref[1]["probs"]) and int(ref[0]["det_count"].sum()) > 0
