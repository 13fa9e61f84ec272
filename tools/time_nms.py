"""Quick NMS timing sweep on the GPU (development aid; bench.py is the contract)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests import synth
from yolov5_obb_amd import nms_rotated_ext

dev = torch.device("cuda:0")
def run(name, dets, scores, thr, iters=5):
    d, s = dets.to(dev), scores.to(dev)
    k = nms_rotated_ext.nms_rotated(d, s, thr)
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); k = nms_rotated_ext.nms_rotated(d, s, thr); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    print(f"{name:28s} n={len(d):7d} thr={thr} kept={len(k):6d}  min {min(ts):8.3f} ms  med {sorted(ts)[len(ts)//2]:8.3f} ms", flush=True)

for n in (1000, 4000, 10000, 30000, 100000):
    run("uniform", *synth.s_uniform(n, 0), 0.4)
    run("clustered", *synth.s_clustered(n, 300, 0), 0.4)
d, s = synth.s_clustered(100000, 3000, 1); run("clustered K=3000", d, s, 0.4)
d, s = synth.s_clustered(100000, 300, 1); d2, _ = synth.with_classes(d, 18, 1); run("clustered +18 classes", d2, s, 0.4)
