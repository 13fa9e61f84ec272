"""The TTA stress tensor (1, 114627, 203) of models/yolo.py:149-161 through the fused driver (conf 0.01, iou 0.4): ms per image.
OBB_NMS_GROUP_AFTER_CUT=1: an image with more than max_nms candidates is grouped by class after the top-30000 cut instead of
falling back to the single list (csrc/nmsobb_impl.h)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests import synth
from yolov5_obb_amd.utils.general import non_max_suppression_obb

dev = torch.device("cuda:0")
p = synth.s_pred(1, 114627, 18, seed=2001, n_obj=300, fg_frac=0.05, device=dev, dtype=torch.float16)
kw = dict(conf_thres=0.01, iou_thres=0.4, multi_label=True, max_det=1500)
for _ in range(5):
    o = non_max_suppression_obb(p, **kw)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    o = non_max_suppression_obb(p, **kw)
e1.record()
torch.cuda.synchronize()
print(f"tta (1,114627,203): {e0.elapsed_time(e1) / 20:.4f} ms per image, {o[0].shape[0]} detections, group_after_cut={os.environ.get('OBB_NMS_GROUP_AFTER_CUT', '0')}")
