import sys, os
os.environ["OBB_NMS_PHASE_PROF"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests import synth
from yolov5_obb_amd import _lib
from yolov5_obb_amd.utils.general import non_max_suppression_obb
import ctypes as C
dev = torch.device("cuda:0")
conf = float(sys.argv[1]) if len(sys.argv) > 1 else 0.001
pred = synth.s_pred(16, 64512, 15, seed=1000, n_obj=120, fg_frac=0.03, device=dev, dtype=torch.float16)
L = _lib.lib()
for _ in range(3):
    out = non_max_suppression_obb(pred, conf_thres=conf, iou_thres=0.45, multi_label=True, max_det=1500)
torch.cuda.synchronize()
L.obb_profile_enable(1)
for _ in range(5):
    out = non_max_suppression_obb(pred, conf_thres=conf, iou_thres=0.45, multi_label=True, max_det=1500)
torch.cuda.synchronize()
ms = (C.c_double * 8)(); cnt = (C.c_int64 * 8)()
L.obb_profile_collect(C.cast(ms, C.c_void_p), C.cast(cnt, C.c_void_p), 8)
print("stages ms:", {n: round(ms[i] / max(1, cnt[i]), 4) for i, n in enumerate(["decode", "segsort", "prep", "nms", "gather"])})
with torch.no_grad():
    objm = pred[..., 4:5] > conf
    cand = (((pred[..., 5:20] * pred[..., 4:5]) > conf) & objm).sum((1, 2))
print("candidates per image", cand.tolist(), "dets", [o.shape[0] for o in out])
