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


def fresh(n, seed):
    """A seeded adversarial mixture on both sides of rule B's envelope (csrc/piou_device.h: |coordinate| <= 70,000, bounding box
    <= 600 x 600): rotated rectangles, slivers, arbitrary four points (bow ties), boxes of 600 .. 1500 px, coordinates up to 2 x 10^5,
    class offsets -- clustered so that real overlaps exist next to the bounding-box-disjoint pairs."""
    g = torch.Generator().manual_seed(seed)
    k = n // 6
    parts = []
    for fam in range(6):
        m = k if fam < 5 else n - 5 * k
        ext = [1024.0, 5000.0, 60000.0, 2000.0, 200000.0, 1024.0][fam]
        nc_ = max(8, m // 25)
        cxy = torch.rand(nc_, 2, generator=g) * ext
        idx = torch.randint(0, nc_, (m,), generator=g)
        xy = cxy[idx] + torch.randn(m, 2, generator=g) * 3
        if fam == 1:
            wh = torch.stack((torch.rand(m, generator=g) * 490 + 10, torch.rand(m, generator=g) * 1.99 + 0.01), 1)
        elif fam == 3:
            wh = torch.rand(m, 2, generator=g) * 900 + 600
        else:
            wh = torch.rand(m, 2, generator=g) * 60 + 4
        th = (torch.rand(m, 1, generator=g) - 0.5) * 3.14159
        q = synth.rbox_to_quad(torch.cat((xy, wh, th), 1))
        if fam == 2:                                               # any four points around the centre
            q = (xy.repeat(1, 4) + (torch.rand(m, 8, generator=g) - 0.5) * 80).float()
        if fam == 5:
            q = q + (torch.randint(0, 18, (m, 1), generator=g).float() * 4096.0)
        parts.append(q)
    quads = torch.cat(parts, 0)
    scores = torch.rand(quads.shape[0], generator=g)
    return quads, synth.tie_free(scores)


def main():
    dev = torch.device("cuda:0")
    if len(sys.argv) > 3 and sys.argv[2] == "fresh":
        quads, s = fresh(30000, int(sys.argv[3]))
        polys = torch.cat([quads, s[:, None]], 1).contiguous().to(dev)
        np.savez(sys.argv[1], **{f"fresh_{thr}": nms_rotated_ext.nms_poly(polys, thr).cpu().numpy() for thr in (0.4, 0.1, 0.02)})
        return
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
    out = {}
    for name, d, s in layouts(n):
        polys = torch.cat([synth.rbox_to_quad(d), s[:, None]], 1).contiguous().to(dev)
        for thr in (0.4, 0.1):
            out[f"{name}_{thr}"] = nms_rotated_ext.nms_poly(polys, thr).cpu().numpy()
    np.savez(sys.argv[1], **out)


if __name__ == "__main__":
    main()
