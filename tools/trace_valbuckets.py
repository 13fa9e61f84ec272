"""Workload for the HIP-API timeline of the in-loop NMS bucket:
    rocprofv3 --hip-trace --kernel-trace --memory-copy-trace -d DIR -o NAME -- python tools/trace_valbuckets.py [loops]
Runs tools/conv_standin.val_buckets `loops` times (default 3) and prints every batch's three buckets, so that a stalled batch
can be found in the trace by its time.  Development aid."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tools import conv_standin
from yolov5_obb_amd import val_sharded

dev = torch.device("cuda:0")
loops = int(sys.argv[1]) if len(sys.argv) > 1 else 3
_run = val_sharded.run
log = []


def run_logged(*a, **k):
    t0 = time.perf_counter_ns()
    r = _run(*a, **k)
    log.append((t0, r.get("dt_batches", [])))
    return r


val_sharded.run = run_logged
for i in range(loops):
    v = conv_standin.val_buckets(dev, n_images=160, batch=16, nc=16, conf_thres=0.25, iou_thres=0.45, half=True, seed=i)
    t0, per = log[-1]
    nms = [round(b[2] * 1e3, 3) for b in per]
    print(f"loop {i}: ms/img pre {v['dt_seconds'][0] / 160 * 1e3:.4f} inf {v['dt_seconds'][1] / 160 * 1e3:.4f} nms {v['dt_seconds'][2] / 160 * 1e3:.4f} | "
          f"stages/batch {v['nms_stages_ms_per_batch']} | passing/img {v['anchors_passing_obj_per_image']}", flush=True)
    print(f"   nms bucket per batch (ms): {nms}   inference per batch (ms): {[round(b[1] * 1e3, 2) for b in per]}", flush=True)
    print(f"   timed loop started at perf_counter_ns {t0}", flush=True)
