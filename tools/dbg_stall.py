"""Development aid: find the rare slow call (tens of ms) of the single-list NMS and print the in-kernel phase timers of that
call (OBB_NMS_PHASE_PROF=1 prints the PREVIOUS call's timers on the next call)."""
import sys, os, time
os.environ["OBB_NMS_PHASE_PROF"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests import synth
from yolov5_obb_amd import nms_rotated_ext
dev = torch.device("cuda:0")
name = sys.argv[1] if len(sys.argv) > 1 else "clustered_k300_18cls"
d, s = synth.regime_100k(name)
dd, ss = d.to(dev), s.to(dev)
for i in range(int(sys.argv[2]) if len(sys.argv) > 2 else 300):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    k = nms_rotated_ext.nms_rotated(dd, ss, 0.4)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3
    print(f"### call {i}: {ms:.2f} ms kept {len(k)}", file=sys.stderr, flush=True)
