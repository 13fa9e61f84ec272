"""Driver for rocprofv3 runs (kernel trace / PMC passes): a calibration copy of known size, then every hot-path
kernel family a few times.  usage: rocprofv3 --kernel-trace [--pmc FETCH_SIZE | --pmc WRITE_SIZE] -d DIR -o NAME -- python tools/prof_kernels.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests import synth
from yolov5_obb_amd import nms_rotated_ext
from yolov5_obb_amd.utils.general import non_max_suppression_obb
from yolov5_obb_amd.utils.loss import ComputeLoss
from yolov5_obb_amd.models.yolo import Detect

dev = torch.device("cuda:0")
REPS = 4
# ---- calibration: 256 MiB float4 copy (read 268435456 B, write 268435456 B per launch)
a = torch.empty(64 * 1024 * 1024, dtype=torch.float32, device=dev).normal_()
b = torch.empty_like(a)
for _ in range(REPS):
    b.copy_(a)
torch.cuda.synchronize()
del a, b
# ---- NMS driver on the shape bench.py's `value` is quoted on (DOTAv1.5: nc = 16, no = 201)
pred = synth.s_pred(16, 64512, 16, seed=1000, n_obj=120, fg_frac=0.03, device=dev, dtype=torch.float16)
for _ in range(REPS):
    out = non_max_suppression_obb(pred, conf_thres=0.25, iou_thres=0.45, multi_label=True, max_det=1500)
torch.cuda.synchronize()
del pred
# ---- rotated NMS at 100k candidates: S-clustered (K = 300), then the same with 18 class offsets (the two regimes bench.py's
#      roofline object chooses between; tools/pmc_json.py splits the dispatches of k_nms_persist<RotGeom, true> in this order)
for regime in ("clustered_k300_raw", "clustered_k300_18cls"):
    d, s = synth.regime_100k(regime)
    d, s = d.to(dev), s.to(dev)
    for _ in range(REPS):
        k = nms_rotated_ext.nms_rotated(d, s, 0.4)
    torch.cuda.synchronize()
# ---- loss forward + backward, BASELINE configs[2] per-GPU shape, fp32
nc = 16
hyp = synth.scaled_hyp(nc, 1024)
p, t = synth.s_loss(16, nc, 1500, 3, imgsz=1024, sizes=[128, 64, 32])
cl = ComputeLoss(synth.FakeModel(nc, hyp, dev))
pg = [x.to(dev).requires_grad_(True) for x in p]
tg = t.to(dev)
for _ in range(REPS):
    for x in pg:
        x.grad = None
    loss, items = cl(pg, tg)
    loss.backward()
torch.cuda.synchronize()
del pg
# ---- Detect inference decode, fp16, bs 16, 1024^2
det = Detect(nc=16, anchors=synth.DEFAULT_ANCHORS, ch=(8, 8, 8))
det.stride = torch.tensor(synth.DEFAULT_STRIDES)
det.anchors /= det.stride.view(-1, 1, 1)
det = det.to(dev).half().eval()
feats = [torch.randn(16, 8, n, n, device=dev, dtype=torch.float16) for n in (128, 64, 32)]
with torch.no_grad():
    for _ in range(REPS):
        z, xs = det([f for f in feats])
torch.cuda.synchronize()
print("done", tuple(z.shape))
