"""Driver for the SQ counter passes (VERDICT r4 item 1 / missing #3): every workload is launched REPS times in a fixed order so
that tools/rocpd_sq.py can attribute the dispatches of a kernel to their workload by launch order.

    rocprofv3 --kernel-trace --pmc <SQ counters> -d DIR -o NAME -- python tools/prof_sq.py

Order (tools/rocpd_sq.py: GROUPS): four 100k regimes of nms_rotated (clustered_k300, clustered_k300_18cls, clustered_k3000,
uniform), the headline step (16, 64512, 201) fp16, the TTA tensor, nms_poly at 30k quads, poly_overlaps 10000 x 1000.
Development aid; bench.py is the contract."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests import synth
from yolov5_obb_amd import nms_rotated_ext, ops
from yolov5_obb_amd.utils.general import non_max_suppression_obb

dev = torch.device("cuda:0")
REPS = int(os.environ.get("SQ_REPS", "3"))
N = 100000


def regimes(n):
    d, s = synth.s_clustered(n, 300, 0); yield "clustered_k300", d, s
    d2, _ = synth.with_classes(d, 18, 0); yield "clustered_k300_18cls", d2, s
    d, s = synth.s_clustered(n, 3000, 0); yield "clustered_k3000", d, s
    d, s = synth.s_uniform(n, 0); yield "uniform", d, s


for name, d, s in regimes(N):
    d, s = d.to(dev), s.to(dev)
    for _ in range(REPS):
        k = nms_rotated_ext.nms_rotated(d, s, 0.4)
    torch.cuda.synchronize()
    print(name, len(k), flush=True)

pred = synth.s_pred(16, 64512, 16, seed=1000, n_obj=120, fg_frac=0.03, device=dev, dtype=torch.float16)
for _ in range(REPS + 1):        # (+ 1: the first call of a shape takes the un-hinted path and is repeated)
    out = non_max_suppression_obb(pred, conf_thres=0.25, iou_thres=0.45, multi_label=True, max_det=1500)
torch.cuda.synchronize()
print("headline", sum(len(o) for o in out), flush=True)
del pred

dq, sq = synth.s_clustered(30000, 300, seed=0)
q9 = torch.cat((synth.rbox_to_quad(dq), sq[:, None]), 1).contiguous().to(dev)
for _ in range(REPS):
    k = nms_rotated_ext.nms_poly(q9, 0.4)
torch.cuda.synchronize()
print("nms_poly_30k", len(k), flush=True)

bo, _ = synth.s_uniform(10000, 3)
qo, _ = synth.s_uniform(1000, 4)
bod, qod = bo.to(dev), qo.to(dev)
for _ in range(REPS):
    m = ops.rbox_overlaps(bod, qod)
torch.cuda.synchronize()
print("poly_overlaps", tuple(m.shape), flush=True)
