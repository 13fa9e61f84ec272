"""One 100k regime through the phase-kernel NMS path a few times (for rocprofv3 --kernel-trace); tools/mk_calls.py prints the
dispatches of the LAST call in order.   python tools/mk_trace.py <regime> [reps] [n]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests import synth
from yolov5_obb_amd import nms_rotated_ext
dev = torch.device("cuda:0")
name = sys.argv[1] if len(sys.argv) > 1 else "clustered_k300"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
n = int(sys.argv[3]) if len(sys.argv) > 3 else 100000
if os.environ.get("MK_CALIB"):          # a copy of known size in front (PMC passes: FETCH_SIZE must read 131072 KB, WRITE_SIZE 262144 KB)
    ca = torch.empty(64 * 1024 * 1024, dtype=torch.float32, device=dev).normal_()
    cb = torch.empty_like(ca)
    cb.copy_(ca); torch.cuda.synchronize()
    del ca, cb
d, s = synth.regime_100k(name, n)
d, s = d.to(dev), s.to(dev)
for _ in range(reps):
    k = nms_rotated_ext.nms_rotated(d, s, 0.4)
    torch.cuda.synchronize()
print(name, len(k))
