"""Loss timing on the GPU (development aid): BASELINE configs[2] per-GPU shape (16,3,{128,64,32}^2,201), nt=1500."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests import synth
from yolov5_obb_amd.utils.loss import ComputeLoss

dev = torch.device("cuda:0")
for dtype in (torch.float32, torch.float16):
    for nt in (0, 50, 1500):
        bs, nc = 16, 16
        hyp = synth.scaled_hyp(nc, 1024)
        p, t = synth.s_loss(bs, nc, nt, 3, imgsz=1024, sizes=[128, 64, 32])
        cl = ComputeLoss(synth.FakeModel(nc, hyp, dev))
        pg = [x.to(device=dev, dtype=dtype).requires_grad_(True) for x in p]
        tg = t.to(dev)
        def step(bwd):
            loss, items = cl(pg, tg)
            if bwd:
                for x in pg: x.grad = None
                loss.backward()
        for bwd in (False, True):
            for _ in range(3): step(bwd)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): step(bwd)
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 20
            nbytes = sum(x.numel() * x.element_size() for x in pg)
            print(f"{str(dtype):14s} nt={nt:5d} {'fwd+bwd' if bwd else 'fwd    '} {ms:8.3f} ms   (head tensors {nbytes/1e6:.0f} MB"
                  f"{', grad write ' + format(nbytes / (ms * 1e-3) / 1e9, '.0f') + ' GB/s equiv' if bwd else ''})", flush=True)
