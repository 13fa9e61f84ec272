"""In-kernel phase times of the NMS kernel inside the headline call (non_max_suppression_obb, bs 16, nc 16)."""
import os, sys
os.environ["OBB_NMS_PHASE_PROF"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tests import synth
from yolov5_obb_amd.utils.general import non_max_suppression_obb
dev = torch.device("cuda:0")
pred = synth.s_pred(16, 64512, 16, seed=1000, n_obj=120, fg_frac=0.03, device=dev, dtype=torch.float16)
kw = dict(conf_thres=0.25, iou_thres=0.45, multi_label=True, max_det=1500)
for i in range(4):
    print("call", i, flush=True)
    out = non_max_suppression_obb(pred, **kw)
    torch.cuda.synchronize()
print([int(o.shape[0]) for o in out])
