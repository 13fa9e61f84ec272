"""Child process of tests/test_nms_gpu.py::test_nms_poly_strict_equals_skip_100k: the quad NMS of three 100k layouts, kept
lists saved to argv[1].  The library reads OBB_NMS_POLY_STRICT once per process, so the parent runs this file twice."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import synth                      # noqa: E402
from yolov5_obb_amd import nms_rotated_ext  # noqa: E402


def layouts(n):
    d, s = synth.s_clustered(n, 300, seed=0)
    s = synth.tie_free(s)
    yield "tile", d, s
    d2 = d.clone(); d2[:, :2] += 5000.0
    yield "shift5000", d2, s
    d3, _ = synth.with_classes(d, 18, 0)
    yield "class_offsets", d3, s


def main():
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
    dev = torch.device("cuda:0")
    out = {}
    for name, d, s in layouts(n):
        polys = torch.cat([synth.rbox_to_quad(d), s[:, None]], 1).contiguous().to(dev)
        for thr in (0.4, 0.1):
            out[f"{name}_{thr}"] = nms_rotated_ext.nms_poly(polys, thr).cpu().numpy()
    np.savez(sys.argv[1], **out)


if __name__ == "__main__":
    main()
