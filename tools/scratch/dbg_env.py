"""Development aid: does the rare ~80 ms stall need this package's kernels at all?  Same loop shape as bench.py's regime loop
(new host tensors -> upload -> timed calls), the timed call being either the NMS or a plain torch kernel of about 1 ms."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests import synth
from yolov5_obb_amd import nms_rotated_ext

dev = torch.device("cuda:0")
mode = os.environ.get("MODE", "torch")          # torch | nms
keep_host = os.environ.get("KEEP_HOST", "0") == "1"
pin = os.environ.get("PIN", "0") == "1"
held = []
a = torch.randn(4096, 4096, device=dev)
T0 = time.perf_counter()
for cycle in range(3):
    for rname in ("clustered_k300_raw", "clustered_k300_18cls", "clustered_k3000", "uniform"):
        d, s = synth.regime_100k(rname, 100000)
        if keep_host:
            held.append((d, s))
        if pin:
            d, s = d.pin_memory(), s.pin_memory()
        dd, ss = d.to(dev), s.to(dev)
        slow, ts = [], []
        for i in range(60):
            t0 = time.perf_counter()
            if mode == "torch":
                b = a @ a
                x = float(b[0, 0].item())
            else:
                nms_rotated_ext.nms_rotated(dd, ss, 0.4)
            dt = (time.perf_counter() - t0) * 1e3
            ts.append(dt)
            if dt > 10:
                slow.append((i, round(dt, 1), round(time.perf_counter() - T0, 2)))
        ts.sort()
        print(cycle, rname, "median", round(ts[len(ts) // 2], 3), "slow:", slow, flush=True)
        del dd, ss
