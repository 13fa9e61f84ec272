"""Stage times (library events) of the fused driver on the TTA stress tensor."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tests import synth
from yolov5_obb_amd import _lib
from yolov5_obb_amd.utils.general import non_max_suppression_obb
dev = torch.device("cuda:0")
L = _lib.lib()
p = synth.s_pred(1, 114627, 18, seed=2001, n_obj=300, fg_frac=0.05, device=dev, dtype=torch.float16)
kw = dict(conf_thres=0.01, iou_thres=0.4, multi_label=True, max_det=1500)
for _ in range(5):
    o = non_max_suppression_obb(p, **kw)
torch.cuda.synchronize()
L.obb_profile_enable(1)
for _ in range(10):
    o = non_max_suppression_obb(p, **kw)
torch.cuda.synchronize()
ms = (C.c_double * 8)(); cnt = (C.c_int64 * 8)()
assert L.obb_profile_collect(C.cast(ms, C.c_void_p), C.cast(cnt, C.c_void_p), 8) == 0
L.obb_profile_enable(0)
names = ["decode", "segsort", "prep", "nms_steps", "gather", "nms_sort", "nms_prep", "s7"]
print({n: round(m / max(c, 1), 4) for n, m, c in zip(names, ms, cnt)}, "detections", o[0].shape[0])
