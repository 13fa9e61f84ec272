"""GPU: the training leg BASELINE configs[2] names, on one device -- the HIP ``ComputeLoss`` inside what train.py wraps around
it: ``DistributedDataParallel`` on the RCCL backend (world size 1: the hooks and gradient buckets are the real ones, the
all-reduce is over one rank), ``torch.cuda.amp.autocast`` (fp16 head outputs) and ``GradScaler`` (a scaled incoming
gradient), followed by an optimizer step (/root/reference/train.py:245,320-345).

Checked: (1) the loss and the gradient that arrives at the head outputs equal the oracle's (pyref.compute_loss on the
fp16-rounded logits, tolerances of tests/test_loss_gpu.py::test_fp16_heads); (2) the parameter gradients DDP hands to the
optimizer are what torch's own autograd produces when the ORACLE's head gradient is pushed through the same graph -- i.e. the
custom autograd.Function composes with autocast, the scaler and DDP's hooks; (3) the scaler does not skip the step, the weights
move, and a second step works (DDP's bucket rebuild after the first iteration)."""
import os

import numpy as np
import pytest
import torch

from oracle import pyref
from tests import synth

pytestmark = pytest.mark.gpu


class TinyObb(torch.nn.Module):
    """Three conv stems + this package's Detect: what ComputeLoss needs from a model (.hyp, .model[-1]) and DDP needs to hook."""

    def __init__(self, nc, hyp):
        super().__init__()
        from yolov5_obb_amd.models.yolo import Detect
        ch = (8, 16, 32)
        self.stems = torch.nn.ModuleList([torch.nn.Conv2d(3, c, 3, stride=s, padding=1) for c, s in zip(ch, (8, 16, 32))])
        det = Detect(nc=nc, anchors=synth.DEFAULT_ANCHORS, ch=ch)
        det.stride = torch.tensor(synth.DEFAULT_STRIDES)
        det.anchors /= det.stride.view(-1, 1, 1)
        self.model = torch.nn.ModuleList([torch.nn.Identity(), det])
        self.hyp = dict(hyp)

    def forward(self, im):
        return self.model[-1]([torch.nn.functional.silu(s(im)) for s in self.stems])


@pytest.fixture()
def nccl_world1(dev):
    import torch.distributed as dist
    if dist.is_initialized():
        pytest.skip("a process group is already up in this process")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29611")
    dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=dev)       # "nccl" is RCCL on ROCm
    yield dist
    dist.destroy_process_group()


def test_ddp_amp_gradscaler_step_with_the_hip_loss(dev, oracle_lib, nccl_world1):
    from torch.nn.parallel import DistributedDataParallel as DDP
    from yolov5_obb_amd.utils.loss import ComputeLoss
    nc, imgsz, bs, nt = 16, 256, 4, 60
    hyp = synth.scaled_hyp(nc, imgsz)
    torch.manual_seed(7)
    model = TinyObb(nc, hyp).to(dev).train()
    ddp = DDP(model, device_ids=[dev.index], output_device=dev.index)
    ddp.hyp = hyp                                                     # train.py:254 hangs the hyper-parameters on the wrapped model
    compute_loss = ComputeLoss(ddp)                                   # (train.py:269 passes the wrapped model as well)
    spec = pyref.LossSpec(hyp, synth.grid_anchors(), torch.tensor(synth.DEFAULT_STRIDES), nc)
    _, targets = synth.s_loss(bs, nc, nt, 11, imgsz=imgsz, sizes=[32, 16, 8])
    g = torch.Generator().manual_seed(3)
    im = torch.rand(bs, 3, imgsz, imgsz, generator=g).to(dev)
    opt = torch.optim.SGD(ddp.parameters(), lr=0.01, momentum=0.9)
    scaler = torch.amp.GradScaler("cuda", init_scale=1024.0)
    params = [p for p in ddp.parameters()]
    before = [p.detach().clone() for p in params]

    with torch.autocast("cuda", dtype=torch.float16):
        pred = ddp(im)                                                # list of (bs, na, ny, nx, no), fp16 under autocast
        for p in pred:
            assert p.dtype == torch.float16
            p.retain_grad()
        loss, items = compute_loss(pred, targets.to(dev))
    scaler.scale(loss).backward()
    scale = scaler.get_scale()

    # (1) loss / head gradients against the oracle on the very logits the head produced
    pc = [p.detach().float().cpu().requires_grad_(True) for p in pred]
    lo, io = pyref.compute_loss(spec, pc, targets.clone())
    lo.backward()
    assert np.allclose(loss.detach().float().cpu().numpy(), lo.detach().numpy(), rtol=2e-3)
    assert np.allclose(items.float().cpu().numpy(), io.numpy(), rtol=2e-3, atol=1e-5)
    for a, b in zip(pred, pc):
        assert a.grad is not None and a.grad.dtype == torch.float16 and torch.isfinite(a.grad).all()
        ref = b.grad * scale
        tol = 2e-3 * ref.abs().max().item() + 1e-4
        assert (a.grad.float().cpu() - ref).abs().max().item() <= tol

    # (2) the parameter gradients behind DDP's hooks = torch autograd of the same graph fed with the ORACLE's head gradient
    got = [p.grad.detach().clone() for p in params]
    assert all(x is not None and torch.isfinite(x).all() for x in got)
    with torch.autocast("cuda", dtype=torch.float16):
        pred2 = model(im)                                             # same weights, no DDP wrapper, no loss kernels
    want = torch.autograd.grad(pred2, [p for p in model.parameters()],
                               grad_outputs=[(b.grad * scale).to(device=dev, dtype=torch.float16) for b in pc], allow_unused=True)
    n_checked = 0
    for x, w in zip(got, want):
        if w is None:
            continue
        tol = 1e-2 * w.float().abs().max().item() + 1e-3           # fp16 head gradients: one fp16 ulp of the largest entries
        assert (x.float() - w.float()).abs().max().item() <= tol
        n_checked += 1
    assert n_checked >= 12                                            # 3 stems + 3 head convs, weights and biases

    # (3) the step is taken, the weights move, and a second iteration works
    scaler.step(opt)
    scaler.update()
    assert scaler.get_scale() == scale                                # no inf / nan found: the step was not skipped
    assert sum(float((p.detach() - b).abs().sum()) for p, b in zip(params, before)) > 0
    opt.zero_grad()
    with torch.autocast("cuda", dtype=torch.float16):
        loss2, _ = compute_loss(ddp(im), targets.to(dev))
    scaler.scale(loss2).backward()
    scaler.step(opt)
    scaler.update()
    assert torch.isfinite(loss2).all() and float(loss2) != float(loss)
