"""Workload for the stalls of the in-loop NMS bucket:
    python tools/trace_valbuckets.py [loops] [freeze]        (or under rocprofv3 --hip-trace --kernel-trace ...)
Runs tools/conv_standin.val_buckets `loops` times (default 3) and prints every batch's three buckets next to (a) the full
(generation-2) passes of CPython's cyclic collector inside the loop (gc.callbacks; `freeze` = 1: gc.freeze() first) and (b) the
cgroup's CFS throttle counters (cpu.stat nr_throttled / throttled_usec) across the loop.  Round 5's finding: the 70-88 ms stalls
are (b) -- 128 OpenMP workers of torch's intra-op pool spinning inside a 16-CPU quota -- not (a).  Development aid."""
import gc
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tools import conv_standin
from yolov5_obb_amd import val_sharded

dev = torch.device("cuda:0")
loops = int(sys.argv[1]) if len(sys.argv) > 1 else 3
freeze = bool(int(sys.argv[2])) if len(sys.argv) > 2 else False
_run = val_sharded.run
log = []
gc_log = []
_t_gc = [0.0]


def _gc_cb(phase, info):
    if phase == "start":
        _t_gc[0] = time.perf_counter()
    elif info.get("generation", 0) == 2:
        gc_log.append((_t_gc[0], time.perf_counter() - _t_gc[0], info.get("collected", 0)))


gc.callbacks.append(_gc_cb)


def run_logged(*a, **k):
    if freeze:
        gc.collect(); gc.freeze()
    t0 = time.perf_counter()
    r = _run(*a, **k)
    log.append((t0, time.perf_counter(), r.get("dt_batches", [])))
    return r


val_sharded.run = run_logged
def throttled():
    try:
        d = dict(l.split() for l in open("/sys/fs/cgroup/cpu.stat"))
        return int(d.get("nr_throttled", 0)), int(d.get("throttled_usec", 0))
    except Exception:
        return 0, 0


print(f"gc.freeze = {freeze}; torch threads {torch.get_num_threads()}; cgroup cpu.max {open('/sys/fs/cgroup/cpu.max').read().strip() if os.path.exists('/sys/fs/cgroup/cpu.max') else '?'}", flush=True)
for i in range(loops):
    th0 = throttled()
    v = conv_standin.val_buckets(dev, n_images=160, batch=16, nc=16, conf_thres=0.25, iou_thres=0.45, half=True, seed=i)
    t0, t1, per = log[-1]
    print(f"loop {i}: ms/img pre {v['dt_seconds'][0] / 160 * 1e3:.4f} inf {v['dt_seconds'][1] / 160 * 1e3:.4f} nms {v['dt_seconds'][2] / 160 * 1e3:.4f} | "
          f"stages/batch {v['nms_stages_ms_per_batch']} | passing/img {v['anchors_passing_obj_per_image']}", flush=True)
    print(f"   per batch (ms): nms {[round(b[2] * 1e3, 2) for b in per]}  inference {[round(b[1] * 1e3, 1) for b in per]}  pre {[round(b[0] * 1e3, 1) for b in per]}", flush=True)
    inside = [(round((s - t0) * 1e3, 1), round(d * 1e3, 1)) for s, d, _ in gc_log if t0 <= s <= t1]
    th1 = throttled()
    print(f"   full collector passes inside the timed loop (at ms, lasted ms): {inside}; loop lasted {(t1 - t0) * 1e3:.1f} ms; "
          f"cgroup throttle events during val_buckets: {th1[0] - th0[0]} ({(th1[1] - th0[1]) / 1e3:.0f} ms of thread time)", flush=True)
