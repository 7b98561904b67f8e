"""Why is the MaskFormer leg slower after another engine ran in the same process?  (bench.py other_configs)"""
import copy, os, sys, time
sys.path.insert(0, os.getcwd())
import torch
import bench
sys.argv = ["bench.py", "--no-cpu-baseline"]
args = bench.parse()
mode = os.environ.get("PROBE", "detr_then_mf")
def mf():
    a = copy.copy(args); a.model, a.family, a.batch, a.size, a.steps, a.warmup = "fai-mf-l-coco-ins", "fai_mf", 16, 800, 10, 3
    r = bench.infer_measure(a, 1, 0, 0, light=True); print(mode, "MF", r["value"], r["ms_per_step"], "mem MB", torch.cuda.memory_allocated() >> 20, torch.cuda.memory_reserved() >> 20, flush=True)
if mode == "mf_only":
    mf(); mf()
elif mode == "detr_then_mf":
    r = bench.infer_measure(args, 1, 0, 0, light=True); print("DETR", r["value"], "mem MB", torch.cuda.memory_allocated() >> 20, torch.cuda.memory_reserved() >> 20)
    mf(); mf()
elif mode == "bf_then_mf":
    a = copy.copy(args); a.model, a.family, a.batch, a.size = "bisenetformer-l-ade", "bisenetformer", 32, 640
    r = bench.infer_measure(a, 1, 0, 0, light=True); print("BF", r["value"])
    mf()
