"""Quad NMS (obb_nms_poly) at the reference's max_nms size and at 100k, S-clustered quads as bench.py's polygon_paths times them.
OBB_NMS_POLY_STRICT=1 in the environment clips every pair (no bounding-box skip)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import synth  # noqa: E402
from yolov5_obb_amd import nms_rotated_ext  # noqa: E402

dev = torch.device("cuda:0")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for n, k, extent in ((30000, 300, 1024.0), (100000, 300, 1024.0), (30000, 3000, 1024.0), (30000, 300, 4096.0)):
    d, s = synth.s_clustered(n, k, seed=0, extent=extent)
    q9 = torch.cat((synth.rbox_to_quad(d), synth.tie_free(s)[:, None]), 1).contiguous().to(dev)
    for thr in (0.4, 0.1):
        kept = nms_rotated_ext.nms_poly(q9, thr)
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(5):
            e0.record()
            kept = nms_rotated_ext.nms_poly(q9, thr)
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1))
        print(f"nms_poly n={n} K={k} extent={extent:.0f} thr={thr}: kept={kept.numel()} min {best:.3f} ms strict={os.environ.get('OBB_NMS_POLY_STRICT', '0')}", flush=True)
