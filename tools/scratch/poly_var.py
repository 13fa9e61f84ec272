"""Per-call host and GPU-event times of nms_poly / nms_rotated at 100k: looking for the source of rare 50 ms calls."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tests import synth
from yolov5_obb_amd import nms_rotated_ext
dev = torch.device("cuda:0")
def times(fn, reps):
    host, gpu = [], []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); t0 = time.perf_counter(); e0.record(); r = fn(); e1.record(); torch.cuda.synchronize()
        host.append(round((time.perf_counter() - t0) * 1e3, 2)); gpu.append(round(e0.elapsed_time(e1), 2))
    return host, gpu, r
dq, sq = synth.s_clustered(100000, 300, seed=0)
q9 = torch.cat((synth.rbox_to_quad(dq), sq[:, None]), 1).contiguous().to(dev)
d, s = dq.to(dev), sq.to(dev)
for rnd in range(2):
    h, g, k = times(lambda: nms_rotated_ext.nms_poly(q9, 0.4), 30)
    print("poly host", h, "\npoly gpu ", g, flush=True)
    h, g, k = times(lambda: nms_rotated_ext.nms_rotated(d, s, 0.4), 30)
    print("rot  host", h, "\nrot  gpu ", g, flush=True)
