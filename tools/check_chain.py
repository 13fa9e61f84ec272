"""Detect decode -> non_max_suppression_obb on the planted-object heads of bench.py's detect_nms_chain, against the CPU oracle
(oracle/pyref.py) run on the very tensor Detect produced.  fp32: exact rows; fp16: rows compared as canonical sets."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from oracle import pyref
from tests import synth
from yolov5_obb_amd.models.yolo import Detect
from yolov5_obb_amd.utils.general import non_max_suppression_obb

dev = torch.device("cuda:0")
bs, nc = 16, 15
kw = dict(conf_thres=0.25, iou_thres=0.45, multi_label=True, max_det=1500)
bad = 0
if os.environ.get("PRELUDE", "0") == "1":          # what bench.py runs before its chain section: same (A, nc) key, other data
    pred = synth.s_pred(bs, 64512, nc, seed=1000, n_obj=120, fg_frac=0.03, device=dev, dtype=torch.float16)
    for _ in range(5):
        non_max_suppression_obb(pred, **kw)
    print("prelude done", flush=True)
for dtype in ((torch.float16,) if os.environ.get("PRELUDE", "0") == "1" else (torch.float32, torch.float16)):
    det = Detect(nc=nc, anchors=synth.DEFAULT_ANCHORS, ch=(8, 8, 8))
    det.stride = torch.tensor(synth.DEFAULT_STRIDES); det.anchors /= det.stride.view(-1, 1, 1)
    det = det.to(dev).to(dtype).eval(); det.m = torch.nn.ModuleList([torch.nn.Identity() for _ in range(3)])
    heads = [h.to(dev) for h in synth.s_head(bs, nc, (128, 64, 32), seed=2000, n_obj=120, dtype=dtype)]
    with torch.no_grad():
        z, _ = det(list(heads))
    got = non_max_suppression_obb(z, **kw)
    for _ in range(3):
        got = non_max_suppression_obb(z, **kw)
    got2 = non_max_suppression_obb(z.clone(), **kw)
    ref = pyref.non_max_suppression_obb(z.cpu().clone(), **kw)
    n_got, n_ref = sum(o.shape[0] for o in got), sum(r.shape[0] for r in ref)
    for b in range(bs):
        g, g2, r = got[b].cpu(), got2[b].cpu(), torch.as_tensor(ref[b])
        same = g.shape == r.shape and (torch.equal(g, r) if dtype == torch.float32 else np.array_equal(synth.canon_rows(g), synth.canon_rows(r)))
        if not same or not torch.equal(g, g2):
            bad += 1
            print(f"  {dtype} image {b}: gpu {tuple(g.shape)} plain-path {tuple(g2.shape)} oracle {tuple(r.shape)}", flush=True)
    print(f"{dtype}: detections gpu {n_got} oracle {n_ref}; images that differ so far: {bad}", flush=True)
print("CHAIN OK" if bad == 0 else "CHAIN MISMATCH")
