"""Val tail per batch of 16 (val.py:209-250) and the headline NMS call with the active binding: python tools/time_valtail.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests import synth
from yolov5_obb_amd import _lib
from yolov5_obb_amd import val as V
from yolov5_obb_amd.utils.general import non_max_suppression_obb

dev = torch.device("cuda:0")
pred = synth.s_pred(16, 64512, 16, seed=1000, n_obj=120, fg_frac=0.03, device=dev, dtype=torch.float16)
kw = dict(conf_thres=0.25, iou_thres=0.45, multi_label=True, max_det=1500)
g = torch.Generator().manual_seed(1)
tg = torch.cat([torch.cat((torch.full((23, 1), float(b)), torch.randint(0, 16, (23, 1), generator=g).float(), torch.rand(23, 2, generator=g) * 1024,
                           torch.rand(23, 1, generator=g) * 100 + 20, torch.rand(23, 1, generator=g) * 20 + 8, (torch.rand(23, 1, generator=g) - 0.5) * 3.14), 1)
                for b in range(16)], 0).to(dev)
shapes = [((1024, 1024), ((1.0, 1.0), (0.0, 0.0)))] * 16
iouv = torch.linspace(0.5, 0.95, 10, device=dev)
import gc
gc.collect(); gc.freeze()
for binding in ("compiled" if _lib.compiled() is not None else "ctypes",):      # OBB_BINDING=ctypes python tools/time_valtail.py for the other one
    for _ in range(20):
        dets = non_max_suppression_obb(pred, **kw)
        V.val_tail_batch(dets, tg, shapes, iouv)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        dets = non_max_suppression_obb(pred, **kw)
    torch.cuda.synchronize()
    t_nms = (time.perf_counter() - t0) / 200 * 1e3
    t0 = time.perf_counter()
    for _ in range(200):
        out = V.val_tail_batch(dets, tg, shapes, iouv)
    torch.cuda.synchronize()
    t_tail = (time.perf_counter() - t0) / 200 * 1e3
    print(f"{binding:9s} non_max_suppression_obb (one tensor, warm) {t_nms:.4f} ms per batch | val_tail_batch {t_tail:.4f} ms per batch of 16 ({sum(len(d) for d in dets)} detections)", flush=True)
