"""poly_overlaps (ops.rbox_overlaps -> obb_rbox_overlaps_f32) and the quad matrix as bench.py's polygon_paths times them."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import synth  # noqa: E402
from yolov5_obb_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for n, k, extent in ((10000, 1000, 1024.0), (10000, 1000, 300.0), (10000, 1000, 4096.0), (20000, 5000, 1024.0)):
    bo, _ = synth.s_uniform(n, 3, extent=extent)
    qo, _ = synth.s_uniform(k, 4, extent=extent)
    bod, qod = bo.to(dev), qo.to(dev)
    r = ops.rbox_overlaps(bod, qod)
    torch.cuda.synchronize()
    ts = []
    for _ in range(10):
        e0.record()
        r = ops.rbox_overlaps(bod, qod)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    print(f"rbox_overlaps {n}x{k} extent={extent:.0f}: median {ts[5]:.3f} ms min {ts[0]:.3f} -> {n * k / ts[5] / 1e-3:.3e} pairs/s  nonzero={int((r != 0).sum())} ({float((r != 0).float().mean()):.4f})", flush=True)
