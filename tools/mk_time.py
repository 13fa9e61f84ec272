"""Wall time of the single-list rotated NMS at 100k per regime and path (development aid).
    python tools/mk_time.py [regimes...]      env: OBB_NMS_MK, OBB_NMS_MK_XLDS as set by the caller"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from tests import synth
from yolov5_obb_amd import nms_rotated_ext
dev = torch.device("cuda:0")
names = sys.argv[1:] or ["clustered_k300", "clustered_k300_18cls", "clustered_k3000", "clustered_k3000_18cls", "uniform"]
for name in names:
    d, s = synth.regime_100k(name)
    d, s = d.to(dev), s.to(dev)
    for _ in range(4):
        k = nms_rotated_ext.nms_rotated(d, s, 0.4)
    torch.cuda.synchronize()
    ts = []
    for _ in range(30):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        k = nms_rotated_ext.nms_rotated(d, s, 0.4)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    print(f"{name:24s} kept {len(k):6d}  median {np.median(ts):.3f} ms  min {np.min(ts):.3f} ms  [MK={os.environ.get('OBB_NMS_MK','-')} XLDS={os.environ.get('OBB_NMS_MK_XLDS','-')}]", flush=True)
