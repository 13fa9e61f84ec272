import sys, os
os.environ["OBB_NMS_PHASE_PROF"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests import synth
from yolov5_obb_amd.utils.general import non_max_suppression_obb
dev = torch.device("cuda:0")
pred = synth.s_pred(16, 64512, 15, seed=1000, n_obj=120, fg_frac=0.03, device=dev, dtype=torch.float16)
for _ in range(4):
    out = non_max_suppression_obb(pred, conf_thres=0.25, iou_thres=0.45, multi_label=True, max_det=1500)
torch.cuda.synchronize()
print([o.shape[0] for o in out])
