"""Stage times of non_max_suppression_obb on the conv stand-in's output (what val_buckets' NMS bucket times)."""
import os, sys, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tools import conv_standin as cs
from yolov5_obb_amd import _lib
from yolov5_obb_amd.utils.general import non_max_suppression_obb
dev = torch.device("cuda:0")
L = _lib.lib()
torch.manual_seed(0)
model = cs.StandinV5s(16).to(dev).eval().half()
loader = cs.SyntheticVal(32, 16, nc=16, seed=0)
im0 = next(iter(loader))[0].to(dev).half() / 255
model.calibrate(im0, 0.25)
with torch.no_grad():
    for _ in range(3):
        out = model(im0)[0]
torch.cuda.synchronize()
print("out", tuple(out.shape), out.dtype, out.is_contiguous(), "objcol attached:", hasattr(out, "_obb_objcol"))
kw = dict(conf_thres=0.25, iou_thres=0.45, multi_label=True)
for _ in range(5):
    o = non_max_suppression_obb(out, **kw)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    o = non_max_suppression_obb(out, **kw)
torch.cuda.synchronize()
print("ms per call (host clock)", (time.perf_counter() - t0) / 20 * 1e3, "detections", [int(x.shape[0]) for x in o][:4])
L.obb_profile_enable(1)
for _ in range(10):
    o = non_max_suppression_obb(out, **kw)
torch.cuda.synchronize()
ms = (C.c_double * 8)(); cnt = (C.c_int64 * 8)()
assert L.obb_profile_collect(C.cast(ms, C.c_void_p), C.cast(cnt, C.c_void_p), 8) == 0
names = ["decode", "segsort", "prep", "nms_steps", "gather", "nms_sort", "nms_prep", "s7"]
print({n: round(m / max(c, 1), 4) for n, m, c in zip(names, ms, cnt)})
x = out.float()
cand = ((x[..., 5:21] * x[..., 4:5]) > 0.25).sum().item()
print("candidates per batch", cand, "anchors passing obj", int((x[..., 4] > 0.25).sum()))
m = (x[..., 4] > 0.25) & ((x[..., 5:21] * x[..., 4:5]) > 0.25).any(-1)
mn = x[..., 2:4].min(-1).values
print("passing anchors per image", m.sum(1).tolist())
print("of them with min side < 1 px:", int((m & (mn < 1.0) & (mn >= 0.001)).sum()), "min side quantiles", torch.quantile(mn[m][:100000], torch.tensor([0.0, 0.01, 0.5, 0.99], device=dev)).tolist())
print("cand per image", ((x[..., 5:21] * x[..., 4:5]) > 0.25).sum((1, 2)).tolist())
from yolov5_obb_amd.utils import general as G
print("meta memo", {k: v for k, v in getattr(G, "_meta_memo", {}).items()} if hasattr(G, "_meta_memo") else None)
