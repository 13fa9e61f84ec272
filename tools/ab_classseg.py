import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests import synth
from yolov5_obb_amd.utils.general import non_max_suppression_obb
dev = torch.device("cuda:0")
pred = synth.s_pred(16, 64512, 15, seed=1000, n_obj=120, fg_frac=0.03, device=dev, dtype=torch.float16)
for conf in (0.25, 0.05, 0.001):
    kw = dict(conf_thres=conf, iou_thres=0.45, multi_label=True, max_det=1500)
    for _ in range(3): out = non_max_suppression_obb(pred, **kw)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): out = non_max_suppression_obb(pred, **kw)
    e1.record(); torch.cuda.synchronize()
    print(f"OBB_NO_CLASS_SEG={os.environ.get('OBB_NO_CLASS_SEG','0')} conf {conf}: {e0.elapsed_time(e1)/10:.3f} ms/batch, dets {sum(o.shape[0] for o in out)}", flush=True)
