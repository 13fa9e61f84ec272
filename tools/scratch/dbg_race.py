"""Development aid: repeated back-to-back NMS calls -- per-call time, abort/retry events, result stability."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from tests import synth
from yolov5_obb_amd import nms_rotated_ext, _lib
import oracle

dev = torch.device("cuda:0")
orig = _lib.retry_on_abort
def noisy(run):
    try:
        return run()
    except _lib.NmsAborted as e:
        print("   ABORT -> retry:", e, flush=True)
        L = _lib.lib(); L.obb_nms_set_max_grid(8)
        try:
            return run()
        finally:
            L.obb_nms_set_max_grid(0)
_lib.retry_on_abort = noisy
for name in sys.argv[1:] or ["clustered_k3000", "uniform_18cls"]:
    d, s = synth.regime_100k(name)
    ref = oracle.nms_rotated(d.numpy(), s.numpy(), 0.4, threads=min(os.cpu_count() or 1, 64))
    dd, ss = d.to(dev), s.to(dev)
    ts, bad = [], 0
    for i in range(30):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        k = nms_rotated_ext.nms_rotated(dd, ss, 0.4)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
        kk = k.cpu().numpy()
        if not np.array_equal(kk, ref):
            bad += 1
            extra = np.setdiff1d(kk, ref); miss = np.setdiff1d(ref, kk)
            print(f"   run {i}: len {len(kk)} vs {len(ref)}  extra {extra[:5]} missing {miss[:5]}", flush=True)
    print(name, "ms:", " ".join(f"{t:.2f}" for t in ts), "| mismatching runs:", bad, flush=True)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        k = nms_rotated_ext.nms_rotated(dd, ss, 0.4)
    e1.record(); torch.cuda.synchronize()
    print(name, "20 back-to-back calls:", e0.elapsed_time(e1) / 20, "ms per call", flush=True)
