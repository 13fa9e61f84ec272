"""CPU: the PRODUCT's loss math header (yolov5_obb_amd/csrc/loss_math.h) compiled with g++ and compared with torch
autograd of the pinned oracle (oracle/pyref.py: bbox_ciou, BCEWithLogits) -- forward values and hand-written
gradients, tolerance 1e-5 (BASELINE.json north_star: "within 1e-5 on IoU/loss scalars")."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest
import torch

from oracle import pyref

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def hl(tmp_path_factory):
    out = tmp_path_factory.mktemp("hl") / "libhostloss.so"
    subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", f"-I{ROOT}/yolov5_obb_amd/csrc",
                    f"{ROOT}/tests/native/host_loss_math.cpp", "-o", str(out), "-lm"], check=True)
    L = C.CDLL(str(out))
    fp = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
    L.hc_ciou.argtypes = [fp, C.c_long, fp]
    L.hc_bce.argtypes = [fp, C.c_long, C.c_float, fp]
    L.hc_focal.argtypes = [fp, C.c_long, C.c_float, C.c_float, fp]
    L.hc_pred.argtypes = [fp, C.c_long, fp]
    L.hc_rem1.argtypes = [fp, C.c_long, fp]
    return L


def test_ciou_forward_and_gradient(hl):
    g = torch.Generator().manual_seed(5)
    n = 20000
    p = torch.cat((torch.rand(n, 2, generator=g) * 1.5 - 0.25, torch.rand(n, 2, generator=g) * 12 + 0.05), 1)
    t = torch.cat((torch.rand(n, 2, generator=g), torch.rand(n, 2, generator=g) * 12 + 0.2), 1)
    p[: n // 4, 2:] = t[: n // 4, 2:] * (1 + 0.05 * torch.randn(n // 4, 2, generator=g))      # near matches
    p = p.clone().requires_grad_(True)
    ciou = pyref.bbox_ciou(p.T, t)
    ciou.sum().backward()
    out = np.zeros((n, 5), np.float32)
    hl.hc_ciou(np.ascontiguousarray(torch.cat((p.detach(), t), 1).numpy()), n, out)
    assert np.allclose(out[:, 0], ciou.detach().numpy(), rtol=1e-5, atol=1e-6)
    gref = p.grad.numpy()
    assert np.allclose(out[:, 1:], gref, rtol=1e-4, atol=2e-6), np.abs(out[:, 1:] - gref).max()


@pytest.mark.parametrize("pw", [1.0, 2.5])
def test_bce_with_logits_forward_and_gradient(hl, pw):
    g = torch.Generator().manual_seed(6)
    n = 20000
    x = (torch.randn(n, generator=g) * 6).requires_grad_(True)
    t = torch.rand(n, generator=g)
    t[: n // 3] = (t[: n // 3] > 0.5).float()
    loss = torch.nn.functional.binary_cross_entropy_with_logits(x, t, pos_weight=torch.tensor([pw]), reduction='none')
    loss.sum().backward()
    out = np.zeros((n, 2), np.float32)
    hl.hc_bce(np.ascontiguousarray(torch.stack((x.detach(), t), 1).numpy()), n, pw, out)
    assert np.allclose(out[:, 0], loss.detach().numpy(), rtol=1e-5, atol=1e-6)
    assert np.allclose(out[:, 1], x.grad.numpy(), rtol=1e-5, atol=1e-6)


def test_pred_box_decode_and_remainder(hl):
    g = torch.Generator().manual_seed(7)
    n = 5000
    lg = (torch.randn(n, 4, generator=g) * 3).requires_grad_(True)
    an = torch.rand(n, 2, generator=g) * 10 + 0.5
    pxy = lg[:, :2].sigmoid() * 2 - 0.5
    pwh = (lg[:, 2:].sigmoid() * 2) ** 2 * an
    torch.cat((pxy, pwh), 1).sum().backward()
    out = np.zeros((n, 8), np.float32)
    hl.hc_pred(np.ascontiguousarray(torch.cat((lg.detach(), an), 1).numpy()), n, out)
    assert np.allclose(out[:, :4], torch.cat((pxy, pwh), 1).detach().numpy(), rtol=1e-5, atol=1e-6)
    assert np.allclose(out[:, 4:], lg.grad.numpy(), rtol=1e-5, atol=1e-6)
    x = ((torch.rand(4000, generator=g) - 0.3) * 200).float()
    r = np.zeros(4000, np.float32)
    hl.hc_rem1(np.ascontiguousarray(x.numpy()), 4000, r)
    assert np.array_equal(r, (x % 1).numpy())


@pytest.mark.parametrize("pw,gamma", [(1.0, 1.5), (2.5, 2.0), (1.0, 0.0)])
def test_focal_bce_forward_and_gradient(hl, pw, gamma):
    """FocalLoss around BCEWithLogitsLoss (utils/loss.py:35-62): element values and hand-written derivative vs torch autograd
    of the restatement; gamma = 0 is the plain BCE."""
    g = torch.Generator().manual_seed(7)
    n = 20000
    x = (torch.randn(n, generator=g) * 5).requires_grad_(True)
    t = torch.rand(n, generator=g)
    t[: n // 3] = (t[: n // 3] > 0.5).float()
    pwt = torch.tensor([pw])
    if gamma > 0:
        loss = pyref.focal_bce(x, t, pwt, gamma) * n          # (mean * n = sum of the element losses)
    else:
        loss = torch.nn.functional.binary_cross_entropy_with_logits(x, t, pos_weight=pwt, reduction='sum')
    loss.backward()
    out = np.zeros((n, 2), np.float32)
    hl.hc_focal(np.ascontiguousarray(torch.stack((x.detach(), t), 1).numpy()), n, pw, gamma, out)
    assert abs(out[:, 0].astype(np.float64).sum() - float(loss)) <= 1e-5 * abs(float(loss))
    assert np.allclose(out[:, 1], x.grad.numpy(), rtol=2e-5, atol=2e-6), np.abs(out[:, 1] - x.grad.numpy()).max()
