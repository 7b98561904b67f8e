"""CPU-only checks: the C-ABI library builds/loads and exports every symbol include/focoos_amd.h declares
(no compute call without a GPU), host logic of the processor / registry / model manager, the product path fails
loudly without its HIP library or a GPU, and the N>1 bench harness (one process per GPU, barrier, max over ranks)
runs under gloo with world_size 2."""
import ctypes
import json
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib_path():
    from focoos_amd import build

    return build.build(force=False, verbose=False)


def header_functions():
    src = open(os.path.join(ROOT, "include", "focoos_amd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(fx_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol(lib_path):
    import torch  # noqa: F401  (HIP runtime first, see focoos_amd/_lib.py)

    lib = ctypes.CDLL(lib_path)
    names = header_functions()
    assert len(names) >= 24
    for n in names:
        assert getattr(lib, n) is not None, n
    from focoos_amd import _lib

    assert sorted(list(_lib.SIGNATURES) + ["fx_error_string"]) == names, "ctypes binding and header drifted apart"
    lib.fx_abi_version.restype = ctypes.c_int
    hdr = int(re.search(r"#define FX_ABI_VERSION (\d+)", open(os.path.join(ROOT, "include", "focoos_amd.h")).read()).group(1))
    assert lib.fx_abi_version() == hdr == _lib.FX_ABI_VERSION   # header, library and ctypes binding agree (a stale .so fails in _lib.load)
    lib.fx_error_string.restype = ctypes.c_char_p
    assert b"invalid argument" in lib.fx_error_string(-1)


def test_fp16_library_exports_the_same_abi_and_reports_its_element_type(lib_path):
    """libfocoos_amd_fp16.so (the same sources, -DFX_FP16=1: training under a loss scale) exports every declared symbol, the same ABI
    version, and fx_build_flags bit 1 tells the two element types apart (the loader checks it: _lib.load)."""
    import torch  # noqa: F401

    from focoos_amd import _lib, build

    p16 = build.build(force=False, verbose=False, fp16=True)
    lib16, lib = ctypes.CDLL(p16), ctypes.CDLL(lib_path)
    for n in header_functions():
        assert getattr(lib16, n) is not None, n
    assert lib16.fx_abi_version() == lib.fx_abi_version() == _lib.FX_ABI_VERSION
    assert (lib16.fx_build_flags() & 2) and not (lib.fx_build_flags() & 2)
    prev = _lib.set_compute_dtype("fp16")
    try:
        assert _lib.lib_path() == p16 and _lib.act_dtype() == torch.float16 and _lib.load().fx_build_flags() & 2
    finally:
        _lib.set_compute_dtype(prev)
    assert _lib.act_dtype() == torch.bfloat16 and not (_lib.load().fx_build_flags() & 2)


def test_loss_scale_state_layout_matches_header(tmp_path):
    """fx_loss_scale_state = {float scale; int32 growth_tracker, good_steps, skipped_steps}: the 16-byte device record FlatAdamW views as
    int32[4] / float32[0]."""
    src = tmp_path / "ls.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "focoos_amd.h"\nint main(void){printf("%zu %zu %zu %zu %zu\\n", sizeof(fx_loss_scale_state), '
                   'offsetof(fx_loss_scale_state, scale), offsetof(fx_loss_scale_state, growth_tracker), offsetof(fx_loss_scale_state, good_steps), '
                   'offsetof(fx_loss_scale_state, skipped_steps));return 0;}\n')
    exe = tmp_path / "ls"
    subprocess.check_call(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    assert list(map(int, subprocess.check_output([str(exe)]).split())) == [16, 0, 4, 8, 12]


def test_conv_desc_layout_matches_header(tmp_path):
    """The ctypes mirror of fx_conv_desc has the layout a C compiler gives the header's struct (plain C, gcc)."""
    from focoos_amd._lib import FxConvDesc

    src = tmp_path / "layout.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "focoos_amd.h"\nint main(void){printf("%zu %zu %zu %zu\\n", sizeof(fx_conv_desc), '
                   'offsetof(fx_conv_desc, y_batch_stride), offsetof(fx_conv_desc, B), offsetof(fx_conv_desc, residual_after_act));return 0;}\n')
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    size, off_ybs, off_b, off_raa = map(int, subprocess.check_output([str(exe)]).split())
    assert ctypes.sizeof(FxConvDesc) == size
    assert FxConvDesc.y_batch_stride.offset == off_ybs and FxConvDesc.B.offset == off_b and FxConvDesc.residual_after_act.offset == off_raa


def test_pw_chain_desc_layout_matches_header(tmp_path):
    """ctypes mirror of fx_pw_chain_desc (grown in ABI 3 by pool / ldp / img_h / img_w) vs the C compiler's layout of the header's struct."""
    from focoos_amd._lib import FxPwChainDesc

    src = tmp_path / "layout2.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "focoos_amd.h"\nint main(void){printf("%zu %zu %zu %zu\\n", sizeof(fx_pw_chain_desc), '
                   'offsetof(fx_pw_chain_desc, act2), offsetof(fx_pw_chain_desc, pool), offsetof(fx_pw_chain_desc, img_w));return 0;}\n')
    exe = tmp_path / "layout2"
    subprocess.check_call(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    size, off_act2, off_pool, off_w = map(int, subprocess.check_output([str(exe)]).split())
    assert ctypes.sizeof(FxPwChainDesc) == size
    assert FxPwChainDesc.act2.offset == off_act2 and FxPwChainDesc.pool.offset == off_pool and FxPwChainDesc.img_w.offset == off_w


def test_missing_library_is_loud(monkeypatch):
    from focoos_amd import _lib

    monkeypatch.setattr(_lib, "_libs", {})
    monkeypatch.setenv("FOCOOS_AMD_LIB", "/nonexistent/libfocoos_amd.so")
    with pytest.raises(_lib.FocoosAmdError, match="no CPU/PyTorch fallback"):
        _lib.load()


@pytest.mark.skipif(torch.cuda.is_available(), reason="CPU-only behaviour")
def test_engine_refuses_to_run_without_gpu():
    from focoos_amd import _lib
    from focoos_amd.model import ModelManager

    with pytest.raises(_lib.FocoosAmdError, match="no CPU fallback"):
        ModelManager.get("fai-detr-l-coco")


def test_registry_and_manager_errors():
    from focoos_amd.model import ModelManager
    from focoos_amd.registry import ModelRegistry

    assert ModelRegistry.exists("fai-detr-l-obj365") and not ModelRegistry.exists("nope")
    with pytest.raises(ValueError):
        ModelRegistry.get_model_info("nope")
    with pytest.raises(ValueError):
        ModelManager.get("nope")
    info = ModelRegistry.get_model_info("fai-detr-l-obj365")
    assert info["config"]["num_classes"] == 365 and len(info["classes"]) == 365


def test_processor_host_logic():
    from focoos_amd.ports import FocoosDetections
    from focoos_amd.processor import DETRProcessor

    p = DETRProcessor({"top_k": 300, "threshold": 0.5}, image_size=640)
    imgs = [np.zeros((480, 600, 3), np.uint8), torch.zeros(3, 100, 50)]
    assert p.get_image_sizes(imgs) == [(480, 600), (100, 50)]
    assert p.get_image_sizes(np.zeros((2, 64, 32, 3), np.uint8)) == [(64, 32)]  # reference quirk H6: one size for a 4-D array
    with pytest.raises(ValueError):
        p.get_image_sizes("x")
    with pytest.raises(ValueError):
        p.train(True).preprocess(imgs, torch.device("cpu"))
    scores = torch.tensor([[0.9, 0.8, 0.1], [0.7, 0.2, 0.1]])
    labels = torch.tensor([[3, 1, 0], [2, 0, 0]], dtype=torch.int32)
    boxes = torch.arange(24, dtype=torch.int32).view(2, 3, 4)
    count = torch.tensor([2, 1], dtype=torch.int32)
    out = DETRProcessor.pack_detections(scores, labels, boxes, count, ["a", "b", "c", "d"])
    assert isinstance(out[0], FocoosDetections) and len(out[0]) == 2 and len(out[1]) == 1
    d = out[0].detections[0]
    assert d.bbox == [0, 1, 2, 3] and d.cls_id == 3 and d.label == "d" and abs(d.conf - 0.9) < 1e-6


def test_bench_two_process_gloo_dry_run():
    """bench.py's N>1 plumbing (RANK/WORLD_SIZE env, barrier, max-over-ranks, single JSON line on rank 0) with gloo."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29671", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "1", "--dry-run"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=240, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] == 5 and j["scaling"] == "weak" and j["value"] > 0


def test_bench_spawns_its_own_ranks():
    """`python bench.py --gpus 2` WITHOUT torchrun starts the two ranks itself (focoos_amd.launch, the mirror of the reference's
    launch(), utils/distributed/dist.py:38-95) and reports n_gpus 2; a mismatch between WORLD_SIZE and --gpus is refused."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    for extra in ([], ["--train"]):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1", "--dry-run"] + extra,
                           capture_output=True, text=True, timeout=240, cwd=ROOT, env=env)
        assert r.returncode == 0, r.stderr[-2000:]
        lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
        assert len(lines) == 1, r.stdout
        j = json.loads(lines[0])
        assert j["n_gpus"] == 2 and j["steps"] == 4 and j["value"] > 0
        if extra:   # --train --dry-run: the per-rank plan of the data-parallel step (VERDICT r3 next #9b)
            plan = j["dp_plan"]
            assert plan["world_size"] == 2 and plan["per_rank_batch"] == 16 and plan["global_batch"] == 32
            assert plan["trainable_parameters"] == 43_361_847 and plan["allreduce_bytes_per_step"] == 4 * 43_361_847   # 173.4 MB fp32 (SURVEY: 173.7)
            assert sum(plan["segment_parameters"].values()) == plan["trainable_parameters"] == sum(plan["bucket_elements"])
            assert max(plan["bucket_elements"]) * 4 <= 64 << 20 and plan["n_buckets"] == 4 and plan["launch_order"][0] == "head"
            assert plan["ring_bytes_sent_per_rank"] == plan["allreduce_bytes_per_step"]          # 2 (N-1)/N = 1 at N = 2
            assert [r["images"] for r in plan["ranks"]] == [[0, 16], [16, 32]]
    r8 = subprocess.run([sys.executable, "-c", "import json; from focoos_amd.train import dp_plan; from focoos_amd.registry import ModelRegistry as R; "
                         "print(json.dumps(dp_plan(R.get_model_info('bisenetformer-l-ade')['config'], 'bisenetformer', 'SyncBN', 8, 8, grad_bytes=2)))"],
                        capture_output=True, text=True, timeout=120, cwd=ROOT, env=env)
    p8 = json.loads(r8.stdout.strip().splitlines()[-1])
    assert p8["global_batch"] == 64 and p8["gradient_dtype"] == "bf16" and 30e6 < p8["allreduce_bytes_per_step"] < 40e6 and len(p8["other_collectives"]) == 2
    assert abs(p8["ring_bytes_sent_per_rank"] - 1.75 * p8["allreduce_bytes_per_step"]) < 8
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--dry-run"], capture_output=True, text=True, timeout=120,
                         cwd=ROOT, env=dict(env, WORLD_SIZE="2", RANK="0"))
    assert bad.returncode != 0 and "WORLD_SIZE=2 but --gpus 1" in (bad.stderr + bad.stdout)


def _launch_probe(path):
    import torch.distributed as dist

    t = torch.tensor([float(dist.get_rank() + 1)])
    dist.all_reduce(t)
    with open(f"{path}.{dist.get_rank()}", "w") as f:
        f.write(f"{os.environ['RANK']} {os.environ['LOCAL_RANK']} {os.environ['WORLD_SIZE']} {dist.get_world_size()} {t.item()}")


def test_launch_mirror(tmp_path):
    """focoos_amd.launch.launch: world 1 runs in-process; world 2 = two spawned ranks with an initialised group and the torchrun env."""
    from focoos_amd.launch import launch

    seen = []
    assert launch(lambda a: seen.append(a) or 7, 1, args=(3,)) == 7 and seen == [3]
    launch(_launch_probe, 2, dist_url="auto", args=(str(tmp_path / "p"),), backend="gloo")
    got = sorted(open(f"{tmp_path}/p.{r}").read() for r in range(2))
    assert got == ["0 0 2 2 3.0", "1 1 2 2 3.0"]
    with pytest.raises(ValueError):
        launch(_launch_probe, 2, num_machines=2, dist_url="auto")


def test_processor_dynamic_axes_match_reference():
    """Processor.get_dynamic_axes of the three mirrors vs the reference processors (fai_detr/processor.py:242-251,
    fai_mf/processor.py:338-345, bisenetformer/processor.py:302-310)."""
    from focoos_amd.processor import BisenetFormerProcessor, DETRProcessor, MaskFormerProcessor
    from focoos_amd.registry import ModelRegistry

    mine = {"fai-detr-l-coco": DETRProcessor(ModelRegistry.get_model_info("fai-detr-l-coco")["config"], 640),
            "fai-mf-l-coco-ins": MaskFormerProcessor(ModelRegistry.get_model_info("fai-mf-l-coco-ins")["config"]),
            "bisenetformer-l-ade": BisenetFormerProcessor(ModelRegistry.get_model_info("bisenetformer-l-ade")["config"])}
    assert mine["fai-detr-l-coco"].get_dynamic_axes().output_names == ["boxes", "logits"]
    assert mine["fai-mf-l-coco-ins"].get_dynamic_axes().dynamic_axes == {"images": {0: "batch", 2: "height", 3: "width"}}
    from oracle import ref_import

    if not ref_import.reference_available():
        return
    builders = {"fai-detr-l-coco": ref_import.build_reference_detr, "fai-mf-l-coco-ins": ref_import.build_reference_mf, "bisenetformer-l-ade": ref_import.build_reference_bf}
    for name, proc in mine.items():
        cfg = ModelRegistry.get_model_info(name)["config"]
        rc = {k: v for k, v in cfg.items() if k != "resolution"} if name.startswith("bisenet") else cfg
        ref = builders[name](rc)[1].get_dynamic_axes()
        got = proc.get_dynamic_axes()
        assert (got.input_names, got.output_names, got.dynamic_axes) == (ref.input_names, ref.output_names, ref.dynamic_axes), name


def test_processor_manager_registry():
    """ProcessorManager (seam B1, processor/processor_manager.py:8-46): lazy family -> processor-class loaders, re-registration wins."""
    from focoos_amd.model import ProcessorManager
    from focoos_amd.processor import BisenetFormerProcessor, DETRProcessor, MaskFormerProcessor
    from focoos_amd.registry import ModelRegistry

    for name, fam, cls in (("fai-detr-l-coco", "fai_detr", DETRProcessor), ("fai-mf-l-coco-ins", "fai_mf", MaskFormerProcessor),
                           ("bisenetformer-l-ade", "bisenetformer", BisenetFormerProcessor)):
        info = ModelRegistry.get_model_info(name)
        p = ProcessorManager.get_processor(fam, info["config"], image_size=info["im_size"])
        assert type(p) is cls
    with pytest.raises(ValueError):
        ProcessorManager.get_processor("unknown_family", {})

    class Custom(DETRProcessor):
        pass

    saved = ProcessorManager._PROCESSOR_MAPPING["fai_detr"]
    try:
        ProcessorManager.register_processor("fai_detr", lambda: Custom)
        assert type(ProcessorManager.get_processor("fai_detr", ModelRegistry.get_model_info("fai-detr-l-coco")["config"], 640)) is Custom
    finally:
        ProcessorManager._PROCESSOR_MAPPING["fai_detr"] = saved


def test_trainer_ports_and_structures():
    """TrainerArgs mirror (focoos/ports.py:970-1065) field-for-field against the reference when it is importable; Boxes / Instances helpers."""
    import dataclasses

    from focoos_amd.ports import Boxes, Instances, TrainerArgs

    a = TrainerArgs(run_name="x")
    assert a.batch_size == 16 and a.learning_rate == 5e-4 and a.clip_gradients == 0.1 and a.backbone_multiplier == 0.1 and a.scheduler == "MULTISTEP"
    from oracle import ref_import

    if ref_import.reference_available():
        ref_import.install()
        from focoos.ports import TrainerArgs as Ref

        ref = {f.name: f.default for f in dataclasses.fields(Ref)}
        mine = {f.name: f.default for f in dataclasses.fields(TrainerArgs)}
        assert set(ref) == set(mine), set(ref) ^ set(mine)
        for k in ref:
            if k in ("output_dir", "num_gpus", "run_name"):
                continue       # environment-dependent defaults (models dir, visible GPU count)
            assert ref[k] == mine[k], (k, ref[k], mine[k])
    b = Boxes(torch.tensor([[0.1, 0.1, 0.5, 0.5], [0.2, 0.2, 0.2, 0.9], [0.9, 0.9, 1.5, 1.2]]))
    b.scale(100, 50)
    b.clip((50, 100))
    inst = Instances((50, 100), boxes=b, scores=torch.tensor([0.9, 0.8, 0.7]), classes=torch.tensor([1, 2, 3]))
    kept = inst[b.nonempty()]
    assert len(kept) == 2 and kept.classes.tolist() == [1, 3] and kept.boxes.tensor[1].tolist() == [90.0, 45.0, 100.0, 50.0]
    with pytest.raises(AssertionError):
        inst.set("bad", torch.zeros(2))


def test_conv_routing_labels(lib_path, monkeypatch):
    """fx_conv2d_variant = the label of the kernel fx_conv2d_nhwc_bf16 routes a descriptor to (one routing function for both; host logic,
    no launch): the production thresholds (flat kernels from 5 000 output pixels, small-M tile up to 16 384) and the error path."""
    import ctypes as C

    from focoos_amd import _lib
    from focoos_amd._lib import FX_ACT, FxConvDesc

    lib = _lib.load()
    for k in ("FX_CONV3_MIN_M", "FX_PW_MIN_M"):
        monkeypatch.delenv(k, raising=False)

    def label(B, H, W, Cc, N, k, stride=1, frag=True, act="relu", pool2=0, out_f32=0):
        d = FxConvDesc()
        d.x = d.w = d.y = d.bias = 0x10000
        d.w_frag = 0x20000 if frag else None
        pad = (k - 1) // 2
        d.B, d.H, d.W, d.C, d.ldx = B, H, W, Cc, Cc
        d.Ho, d.Wo = ((H + 1) // 2, (W + 1) // 2) if pool2 else ((H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1)
        d.N, d.ldy, d.ldr = N, (N + 7) // 8 * 8, 0
        d.KH, d.KW, d.stride, d.pad, d.pool2, d.act, d.out_f32 = k, k, stride, pad, pool2, FX_ACT[act], out_f32
        buf = C.create_string_buffer(64)
        rc = lib.fx_conv2d_variant(C.byref(d), buf, 64)
        return buf.value.decode() if rc == 0 else rc

    assert label(16, 40, 40, 256, 256, 3) == "conv3x3_kplane<256>"            # M = 25 600: a 16-image part's 40x40 level
    assert label(8, 40, 40, 256, 256, 3) == "conv3x3_kplane<256>"               # M = 12 800 >= 5 000
    assert label(2, 40, 40, 256, 256, 3) == "conv_igemm<128,128,64>"          # M = 3 200 < 5 000
    assert label(16, 80, 80, 256, 256, 3, frag=False) == "conv_igemm_dma<256,256>"   # no fragment copy: implicit GEMM (DMA tiles from 40 000 pixels)
    assert label(16, 40, 40, 256, 256, 3, out_f32=1).startswith("conv_igemm")   # fp32 pre-BatchNorm output: not on the halo kernel
    assert label(16, 40, 40, 1024, 256, 1) == "pw_flat<K1024>"
    assert label(16, 20, 20, 512, 512, 3) == "conv3x3_kplane<512>"               # res5 branch2b of a 16-image part: two 256-channel tiles per pixel tile
    assert label(16, 20, 20, 512, 2048, 1) == "pw_kplane<K512>"                  # M = 6400 >= 5000: res5 branch2c of a 16-image part
    assert label(4, 20, 20, 256, 1024, 1) == "conv_igemm<64,64,256,1stage>"    # small M, K <= 1024
    assert label(16, 160, 160, 64, 256, 1, frag=False) == "conv_igemm<128,128,32>"
    assert label(16, 320, 320, 32, 32, 3, frag=False) == "conv_igemm<128,32,32>"
    assert label(16, 160, 160, 256, 512, 1, pool2=1, frag=False) == "conv_igemm<128,128,64,pool>"
    assert label(16, 40, 40, 250, 256, 3) == -1                                # C % 32 != 0: FX_ERR_INVALID_ARG, as the launch would return
    monkeypatch.setenv("FX_CONV3_MIN_M", "0")
    assert label(2, 16, 16, 64, 64, 3) == "conv3x3_kplane<64>"                   # the kernel tests' lowered threshold


def test_pack_entry_blocks(lib_path):
    """fx_pack_entry_blocks = workgroups of one fx_pack_weights_many_f32 table entry: tiles of 8 output channels x at most 2304
    (input channel, tap) elements, the channel chunk a multiple of 8 (host logic; the Python packer sums these into first_block)."""
    from focoos_amd import _lib

    lib = _lib.load()
    assert lib.fx_pack_entry_blocks(256, 256, 3, 3) == 32 * 1          # 256 x 9 = 2304: one chunk
    assert lib.fx_pack_entry_blocks(512, 512, 3, 3) == 64 * 2          # two chunks of 256 channels
    assert lib.fx_pack_entry_blocks(2048, 512, 1, 1) == 256 * 1        # pointwise: up to 2304 channels per tile
    assert lib.fx_pack_entry_blocks(365, 256, 1, 1) == 46              # ragged last tile of 5 output channels
    assert lib.fx_pack_entry_blocks(4, 4, 1, 1) == 1
    assert lib.fx_pack_entry_blocks(64, 3000, 1, 1) == 8 * 2
    assert lib.fx_pack_entry_blocks(64, 64, 17, 17) == -1              # 289 taps: more than a tile holds 8 channels of
    assert lib.fx_pack_entry_blocks(0, 64, 1, 1) == -1


def test_msda_slab_backward_support_predicate():
    """fx_msda_bwd_slab_supported (host logic of csrc/train_ops.hip): RT-DETR's decoder shapes are covered, a level wider than a slab
    or more taps than the LDS holds are not (the caller then keeps the fp32-atomic backward)."""
    import ctypes as C

    import numpy as np

    from focoos_amd import _lib

    lib = _lib.load()

    def ok(shapes, P, Q, bf16=1, M=8):
        a = np.ascontiguousarray(np.array(shapes, dtype=np.int32).reshape(-1))
        return lib.fx_msda_bwd_slab_supported(a.ctypes.data, len(shapes), P, Q, M, bf16)

    assert ok([[80, 80], [40, 40], [20, 20]], 4, 300) == 1
    assert ok([[80, 80], [40, 40], [20, 20]], 4, 300, bf16=0) == 1
    assert ok([[80, 80], [40, 40], [20, 20]], 4, 500) == 1
    assert ok([[100, 100], [50, 50], [25, 25]], 4, 13125) == 0      # the mask families' pixel-decoder encoder: every pixel is a query
    assert ok([[2, 4000]], 4, 10) == 0
    assert ok([[8, 8]], 4, 10, M=4) == 0


def test_adapter_refuses_inputs_without_a_plan_instead_of_running_the_stock_graph():
    """VERDICT r3 / DESIGN §7.1: the integration adapters never execute the reference's own PyTorch graph - sizes below 32 and gradients
    w.r.t. the images raise; every family accepts any size >= 32 (ceil-size engines; RT-DETR since round 5: ADVICE r4, ragged training
    batches are padded to the batch maximum, which the reference accepts whatever it is)."""
    from focoos_amd import _lib
    from focoos_amd.integration import _require_engine_input

    _require_engine_input(torch.zeros(1, 3, 64, 96))
    _require_engine_input(torch.zeros(2, 64, 96, 3, dtype=torch.uint8))
    _require_engine_input(torch.zeros(1, 3, 600, 800))
    _require_engine_input(torch.zeros(1, 750, 500, 3))
    with pytest.raises(_lib.FocoosAmdError, match="multiple of 32"):   # an explicit granularity is still honoured
        _require_engine_input(torch.zeros(1, 750, 512, 3), 32)
    with pytest.raises(_lib.FocoosAmdError, match="smaller than 32"):
        _require_engine_input(torch.zeros(1, 3, 20, 800))
    with pytest.raises(_lib.FocoosAmdError, match="input images"):
        _require_engine_input(torch.zeros(1, 3, 64, 64, requires_grad=True))


def test_model_manager_resolution_order_and_local_run_directories(tmp_path, monkeypatch):
    """ModelManager.get resolves like the reference (model_manager.py:74-91, 158-188; the cases of its tests/test_model_manager.py): an
    explicit ModelInfo, a registry name, else a local run directory holding model_info.json - as given or under MODELS_DIR - whose bare
    ``model_final.pth`` is resolved against the directory; ModelRegistry.get_model_info also takes the path of such a JSON
    (model_registry.py:50-71).  A stub family stands in for the engine classes: no GPU in this test."""
    import json

    from focoos_amd import model as Mo
    from focoos_amd.model import ModelManager
    from focoos_amd.ports import ModelInfo
    from focoos_amd.registry import ModelRegistry

    built = []

    class StubNN:
        family, device, dtype = "fai_detr", torch.device("cpu"), torch.float32

        def __init__(self, cfg, device="cpu", seed=0):
            self.cfg, self.loaded = cfg, None
            built.append(self)

        def eval(self):
            return self

        def load_state_dict(self, state, strict=False):
            self.loaded = state

    monkeypatch.setitem(ModelManager._MODEL_MAPPING, "fai_detr", lambda: StubNN)
    base = ModelRegistry.get_model_info("fai-detr-l-coco")
    run = tmp_path / "my_run"
    run.mkdir()
    info = {k: base[k] for k in ("model_family", "classes", "im_size", "task", "config")}
    info.update(name="my_run", weights_uri="model_final.pth", status="TRAINING_COMPLETED", train_args={"max_iters": 10}, focoos_version="0.25.0")
    (run / "model_info.json").write_text(json.dumps(info))
    w = {"model": {"a": torch.ones(2)}, "iteration": 10}
    torch.save(w, run / "model_final.pth")
    # a run directory given by path; fields the engine does not know are dropped, the bare weights name is resolved
    import warnings as W

    with W.catch_warnings(record=True) as rec:
        W.simplefilter("always")
        fm = ModelManager.get(str(run))
    assert not [r for r in rec if "no pretrained weights" in str(r.message)]
    assert fm.model_info.name == "my_run" and fm.model_info.weights_uri == str(run / "model_final.pth")
    assert torch.equal(built[-1].loaded["a"], torch.ones(2))                     # the "model" entry of the checkpoint
    # ... by name under MODELS_DIR, with config overrides (dict and kwargs) on top of the run's own config
    monkeypatch.setattr(Mo, "MODELS_DIR", str(tmp_path))
    fm = ModelManager.get("my_run", config={"threshold": 0.25}, top_k=100)
    assert built[-1].cfg["threshold"] == 0.25 and built[-1].cfg["top_k"] == 100 and built[-1].cfg["num_classes"] == 80
    # a pathlib.Path works like a string
    assert ModelManager.get(run).model_info.name == "my_run"
    # the registry by JSON path
    d = ModelRegistry.get_model_info(str(run / "model_info.json"))
    assert d["name"] == "my_run" and d["config"] == base["config"] and d["description"] is None
    assert ModelInfo.from_json(str(run / "model_info.json")).im_size == base["im_size"]
    # error cases, with the reference's messages
    with pytest.raises(ValueError, match="not exists"):
        ModelManager.get("no_such_run")
    (tmp_path / "empty_run").mkdir()
    with pytest.raises(ValueError, match="Model info not found"):
        ModelManager.get("empty_run")
    with pytest.raises(ValueError, match="⚠️ Model /nonexistent/model.json not found"):
        ModelRegistry.get_model_info("/nonexistent/model.json")
    (tmp_path / "bad.json").write_text(json.dumps({"name": "x"}))
    with pytest.raises(ValueError, match="required field"):
        ModelRegistry.get_model_info(str(tmp_path / "bad.json"))
    with pytest.raises(NotImplementedError, match="hub://"):
        ModelManager.get("hub://user/ref")
    missing = dict(info, weights_uri=str(run / "gone.pth"))
    with pytest.raises(FileNotFoundError):
        ModelManager.get("x", model_info=ModelInfo.from_json(missing))
    with pytest.raises(ValueError, match="not supported"):
        ModelManager.get("x", model_info=ModelInfo.from_json(dict(info, model_family="rtmo")))


def test_dp_plan_counts_the_syncbn_collectives_of_the_real_module_tree():
    """the per-step SyncBN collective count stated by dp_plan (RT-DETR-L: 97 BatchNorm layers -> 194 small all-reduces,
    154 if the 20 sibling pairs shared theirs) equals what the REAL trainable module tree would issue: one statistics all-reduce in the
    forward and one in the backward per batch-statistics layer (train_nn._bn_forward / _bn_backward)."""
    from unittest import mock

    from focoos_amd.registry import ModelRegistry
    from focoos_amd.train import dp_plan

    cfg = ModelRegistry.get_model_info("fai-detr-l-obj365")["config"]
    plan = dp_plan(cfg, "fai_detr", "SyncBN", 8, 16)
    sb = plan["syncbn"]
    assert sb["batchnorm_layers"] == 97 and sb["collectives_per_step"] == 194
    assert sb["sibling_pairs_that_could_share_a_collective"] == 20 and sb["collectives_per_step_with_siblings_coalesced"] == 154
    # as built (round 6, train_nn._SiblingConvBnFn): all 20 pairs share the forward collective, the 8 shortcut / CSP pairs the backward one too
    assert sb["forward_collectives_as_built"] == 77 and sb["backward_collectives_as_built"] == 89 and sb["collectives_per_step_as_built"] == 166
    assert dp_plan(cfg, "fai_detr", "FrozenBN", 8, 16)["syncbn"] is None
    with mock.patch("torch.cuda.is_available", return_value=True), mock.patch("focoos_amd._lib.load", return_value=None):
        from focoos_amd.train_detr import FAIDetrTrainable

        net = FAIDetrTrainable(cfg, norm="SyncBN")
    layers = [m for m in net.modules() if getattr(m, "norm_mode", None) == "SyncBN" and hasattr(m, "_norm_h")]
    live = [m for m in layers if m._norm_h.weight.requires_grad]
    # the dead mask_features conv of RT-DETR is frozen out of the training graph (train_nn.HybridEncoder): every other BatchNorm is live
    assert len(layers) >= 97 and len(live) in (97, 96), (len(layers), len(live))


def test_binary_mask_to_base64_decodes_to_the_mask_with_cv2_like_framing():
    """focoos/utils/vision.py:270-293.  The standalone surface GUARANTEES decode-equality: the base64 string is a PNG that
    decodes (PIL) to the 0 / 255 image of the mask - 1x1, odd sizes, a mask whose compressed stream spans several 8 192-byte IDAT chunks.  The
    framing follows cv2.imencode's defaults (SUB filter on every row, zlib level 1 / Z_RLE, 8-bit grayscale, IHDR + IDAT... + IEND only);
    byte-equality with cv2 cannot be pinned here (no cv2; the reference's own test computes its expectation with cv2 at run time)."""
    import base64
    import io
    import struct
    import zlib

    from PIL import Image

    from focoos_amd.processor import binary_mask_to_base64

    rng = np.random.RandomState(0)
    for shape in [(2, 2), (1, 1), (5, 7), (64, 64), (300, 517)]:
        m = rng.rand(*shape) > 0.5
        raw = base64.b64decode(binary_mask_to_base64(m))
        im = np.array(Image.open(io.BytesIO(raw)))
        assert im.dtype == np.uint8 and np.array_equal(im, m.astype(np.uint8) * 255), shape
        # container: signature, IHDR (8-bit gray, no interlace), IDAT chunks of at most 8 192 bytes, IEND; every row filtered with SUB (type 1)
        assert raw[:8] == b"\x89PNG\r\n\x1a\n"
        pos, tags, idat = 8, [], b""
        while pos < len(raw):
            n, tag = struct.unpack(">I", raw[pos:pos + 4])[0], raw[pos + 4:pos + 8]
            body = raw[pos + 8:pos + 8 + n]
            assert struct.unpack(">I", raw[pos + 8 + n:pos + 12 + n])[0] == zlib.crc32(tag + body) & 0xFFFFFFFF
            tags.append((tag, n))
            if tag == b"IDAT":
                idat += body
            if tag == b"IHDR":
                assert struct.unpack(">IIBBBBB", body) == (shape[1], shape[0], 8, 0, 0, 0, 0)
            pos += 12 + n
        assert tags[0][0] == b"IHDR" and tags[-1] == (b"IEND", 0) and all(t == b"IDAT" and n <= 8192 for t, n in tags[1:-1])
        if shape == (300, 517):
            assert len(tags) > 4          # several IDAT chunks
        rows = zlib.decompress(idat)
        assert len(rows) == shape[0] * (shape[1] + 1) and set(rows[:: shape[1] + 1]) == {1}
    # the reference's own fixture mask (tests/utils/conftest.py:12-15)
    s = binary_mask_to_base64(np.array([[1, 0], [0, 1]], dtype=bool))
    assert np.array_equal(np.array(Image.open(io.BytesIO(base64.b64decode(s)))), np.array([[255, 0], [0, 255]], dtype=np.uint8))
