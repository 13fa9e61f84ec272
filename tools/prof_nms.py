import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests import synth
from yolov5_obb_amd import nms_rotated_ext
dev = torch.device("cuda:0")
kind = sys.argv[1] if len(sys.argv) > 1 else "clustered"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
d, s = (synth.s_clustered(n, 300, 0) if kind == "clustered" else synth.s_uniform(n, 0))
d, s = d.to(dev), s.to(dev)
for _ in range(5):
    k = nms_rotated_ext.nms_rotated(d, s, 0.4)
torch.cuda.synchronize()
print(kind, n, "kept", len(k))
