"""Stage split (library events) of the fused driver on the secondary bench shapes (development aid)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests import synth
from yolov5_obb_amd import _lib
from yolov5_obb_amd.utils.general import non_max_suppression_obb
dev = torch.device("cuda:0"); L = _lib.lib()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
cases = {"nc2": (synth.s_pred(16, 64512, 2, seed=2002, n_obj=120, fg_frac=0.03, device=dev, dtype=torch.float16), dict(conf_thres=0.25, iou_thres=0.45, multi_label=True, max_det=1500)),
         "tta": (synth.s_pred(1, 114627, 18, seed=2001, n_obj=300, fg_frac=0.05, device=dev, dtype=torch.float16), dict(conf_thres=0.01, iou_thres=0.4, multi_label=True, max_det=1500)),
         "dense": (synth.s_pred(16, 64512, 16, seed=2003, n_obj=4000, fg_frac=0.08, device=dev, dtype=torch.float16), dict(conf_thres=0.25, iou_thres=0.45, multi_label=True, max_det=1500))}
for name, (p, kw) in cases.items():
    if len(sys.argv) > 1 and name not in sys.argv[1:]: continue
    for _ in range(4): o = non_max_suppression_obb(p, **kw)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(20): o = non_max_suppression_obb(p, **kw)
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 20
    L.obb_profile_enable(1)
    for _ in range(10): o = non_max_suppression_obb(p, **kw)
    ms = (C.c_double * 8)(); cnt = (C.c_int64 * 8)()
    L.obb_profile_collect(C.cast(ms, C.c_void_p), C.cast(cnt, C.c_void_p), 8); L.obb_profile_enable(0)
    st = {n: round(ms[i] / max(1, cnt[i]), 4) for i, n in enumerate(("decode", "sort", "prep", "nms", "gather"))}
    print(f"{name}: {t:.4f} ms per call, stages {st}, rows {sum(int(x.shape[0]) for x in o)}", flush=True)
