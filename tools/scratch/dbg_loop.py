import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests import synth
from yolov5_obb_amd import nms_rotated_ext
dev = torch.device("cuda:0")
d, s = synth.regime_100k(sys.argv[1] if len(sys.argv) > 1 else "clustered_k300_18cls")
dd, ss = d.to(dev), s.to(dev)
ts = []
for i in range(200):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); k = nms_rotated_ext.nms_rotated(dd, ss, 0.4); e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1))
print("max", max(ts), "at", ts.index(max(ts)), "median", sorted(ts)[100], "slow calls", [(i, round(t, 1)) for i, t in enumerate(ts) if t > 5])
