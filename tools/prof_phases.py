"""In-kernel phase times of the persistent NMS kernel (OBB_NMS_PHASE_PROF=1): development aid."""
import sys, os
os.environ["OBB_NMS_PHASE_PROF"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests import synth
from yolov5_obb_amd import nms_rotated_ext
dev = torch.device("cuda:0")
cases = [("uniform", 1000), ("clustered", 1000), ("uniform", 10000), ("clustered", 10000), ("uniform", 100000), ("clustered", 100000)]
for kind, n in cases:
    d, s = (synth.s_clustered(n, 300, 0) if kind == "clustered" else synth.s_uniform(n, 0))
    d, s = d.to(dev), s.to(dev)
    print(kind, n, flush=True)
    for _ in range(3):
        k = nms_rotated_ext.nms_rotated(d, s, 0.4)
    torch.cuda.synchronize()
