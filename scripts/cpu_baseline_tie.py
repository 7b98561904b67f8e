"""Ties the CPU baseline's two kinds together on the build box: the REAL reference (FAIDetr + DETRProcessor imported from
/root/reference through oracle/ref_import) and the oracle port (oracle/detr_oracle.py) timed in ONE process on the same images, weights and
threads - bench.py's `cpu_baseline` is kind "port" on the GPU box (no reference tree there) and quotes this file as `reference_tie`.
    python scripts/cpu_baseline_tie.py > profiles/r06_cpu_baseline_tie.json"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--threads", type=int, default=8)
ap.add_argument("--iters", type=int, default=3)
a = ap.parse_args()
base = dict(model="fai-detr-l-obj365", family="fai_detr", size=640, cpu_batch=1, cpu_iters=a.iters, cpu_threads=a.threads)
ref = bench.cpu_baseline(argparse.Namespace(**base))
port = bench.cpu_baseline(argparse.Namespace(cpu_force_port=True, **base))
assert ref["kind"] == "reference" and port["kind"] == "port", (ref["kind"], port["kind"])
print(json.dumps({"reference_images_per_s": ref["bs1_images_per_s"], "port_images_per_s": port["bs1_images_per_s"],
                  "port_over_reference": round(port["bs1_images_per_s"] / ref["bs1_images_per_s"], 3), "threads": ref["cores"], "host_cores": os.cpu_count(),
                  "where": "build container (8 cores), fai-detr-l-obj365 bs=1 640x640, seed-0 weights, median of %d passes after a warm-up" % a.iters,
                  "reference": ref, "port": port}, indent=1))
