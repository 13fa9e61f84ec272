import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests import synth
from yolov5_obb_amd.utils import general
from yolov5_obb_amd import nms_rotated_ext
dev = torch.device("cuda:0")
pred = synth.s_pred(2, 3000, 15, seed=5).to(dev)
for rep in range(3):
    out = general.non_max_suppression_obb(pred, conf_thres=0.25, iou_thres=0.45, multi_label=True)
    print("small call", rep, [o.shape[0] for o in out], dict(general._cand_memo))
pred = synth.s_pred(16, 64512, 15, seed=1000, n_obj=120, fg_frac=0.03, device=dev, dtype=torch.float16)
for rep in range(3):
    out = general.non_max_suppression_obb(pred, conf_thres=0.25, iou_thres=0.45, multi_label=True, max_det=1500)
    print("bench call", rep, [o.shape[0] for o in out])
d, s = synth.s_clustered(5000, 90, 1)
print("plain nms", nms_rotated_ext.nms_rotated(d.to(dev), s.to(dev), 0.4).numel())
