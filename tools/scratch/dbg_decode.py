"""Development aid: k_decode time (library stage events) on a Detect-produced tensor: cold (right after Detect wrote it) or
warm (second call on the same tensor), with / without the objectness column, with a threshold that passes nothing."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import torch
from tests import synth
from yolov5_obb_amd import _lib
from yolov5_obb_amd.models.yolo import Detect
from yolov5_obb_amd.utils.general import non_max_suppression_obb

dev = torch.device("cuda:0")
L = _lib.lib()
bs, nc = 16, 15
def collect():
    ms = (C.c_double * 8)(); cnt = (C.c_int64 * 8)()
    L.obb_profile_collect(C.cast(ms, C.c_void_p), C.cast(cnt, C.c_void_p), 8)
    return [m / max(1, c) for m, c in zip(ms, cnt)]
det = Detect(nc=nc, anchors=synth.DEFAULT_ANCHORS, ch=(8, 8, 8))
det.stride = torch.tensor(synth.DEFAULT_STRIDES); det.anchors /= det.stride.view(-1, 1, 1)
det = det.to(dev).half().eval()
det.m = torch.nn.ModuleList([torch.nn.Identity() for _ in range(3)])
heads = [h.to(dev) for h in synth.s_head(bs, nc, (128, 64, 32), seed=2000, n_obj=120, dtype=torch.float16)]
pred = synth.s_pred(bs, 64512, nc, seed=1000, n_obj=120, fg_frac=0.03, device=dev, dtype=torch.float16)
trash = torch.empty(1 << 28, dtype=torch.float32, device=dev)
for conf in (0.25, 0.9999):
    kw = dict(conf_thres=conf, iou_thres=0.45, multi_label=True, max_det=1500)
    for couple in (False, True):
        det.couple_nms = couple
        for mode in ("cold: Detect, then NMS", "warm: NMS again on the same z", "cold: 1 GiB memset, then NMS"):
            with torch.no_grad():
                z, _ = det(list(heads))
            for _ in range(5):
                non_max_suppression_obb(z, **kw)
            L.obb_profile_enable(1)
            for _ in range(20):
                if mode.startswith("cold: Detect"):
                    with torch.no_grad():
                        z, _ = det(list(heads))
                elif mode.startswith("cold: 1"):
                    trash.zero_()
                non_max_suppression_obb(z, **kw)
            torch.cuda.synchronize()
            st = collect(); L.obb_profile_enable(0)
            print(f"conf {conf} column {couple!s:5} {mode:32} k_decode {st[0]*1e3:7.1f} us", flush=True)
kw = dict(conf_thres=0.25, iou_thres=0.45, multi_label=True, max_det=1500)
for mode in ("warm", "cold: 1 GiB memset"):
    for _ in range(5):
        non_max_suppression_obb(pred, **kw)
    L.obb_profile_enable(1)
    for _ in range(20):
        if mode != "warm":
            trash.zero_()
        non_max_suppression_obb(pred, **kw)
    torch.cuda.synchronize()
    st = collect(); L.obb_profile_enable(0)
    print(f"s_pred (bench tensor) {mode:20} k_decode {st[0]*1e3:7.1f} us", flush=True)
