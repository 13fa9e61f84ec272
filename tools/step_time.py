"""The headline step (bench.py's tensors: four rotated (16, 64512, 201) fp16 tensors) timed in windows of 200 calls, plus a
checksum of the outputs (development aid; A/B switches are read from the environment by the library).
    python tools/step_time.py [windows]"""
import os, sys, time, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from tests import synth
from yolov5_obb_amd.utils.general import non_max_suppression_obb
dev = torch.device("cuda:0")
bs, A, nc = int(os.environ.get("BS", "16")), 64512, 16
kw = dict(conf_thres=0.25, iou_thres=0.45, multi_label=True, max_det=1500)
preds = [synth.s_pred(bs, A, nc, seed=1000 + r, n_obj=120, fg_frac=0.03, device=dev, dtype=torch.float16) for r in range(4)]
torch.cuda.synchronize()
h = hashlib.sha256()
for p in preds:
    for _ in range(3):
        out = non_max_suppression_obb(p, **kw)
    for o in out:
        h.update(o.cpu().numpy().tobytes())
t_spin = time.perf_counter()
i = 0
while time.perf_counter() - t_spin < 0.4:
    out = non_max_suppression_obb(preds[i % 4], **kw); i += 1
torch.cuda.synchronize()
ws = []
for w in range(int(sys.argv[1]) if len(sys.argv) > 1 else 5):
    t0 = time.perf_counter()
    for i in range(200):
        out = non_max_suppression_obb(preds[i % 4], **kw)
    torch.cuda.synchronize()
    ws.append((time.perf_counter() - t0) / 200 * 1e3)
print(f"step median {np.median(ws):.4f} ms  windows {[round(w, 4) for w in ws]}  rows {sum(int(o.shape[0]) for o in out)}  sha {h.hexdigest()[:16]}  "
      f"[HELPERS={os.environ.get('OBB_NMS_SMALL_HELPERS', '-')} BS={bs}]", flush=True)
