"""GPU parity of the Detect inference decode, the CSL encode and rbox2poly / poly2hbb kernels (csrc/head.hip)
against the CPU oracle (oracle/pyref.py, pinned to the reference's models/yolo.py and utils/rboxs_utils.py) and
against the fixtures frozen from the reference itself (tests/golden).

Tolerances: the permuted raw head is a copy -> bit-exact.  Decoded values go through sigmoid/exp, whose last bit
differs between libm implementations: fp32 within 2e-6 relative (north_star: 1e-5), fp16 within one fp16 ulp."""
import os

import numpy as np
import pytest
import torch

from oracle import pyref
from tests import synth

pytestmark = pytest.mark.gpu
G = np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_outputs.npz"))


def make_detect(nc, ch, dev, dtype=torch.float32, seed=0):
    from yolov5_obb_amd.models.yolo import Detect
    torch.manual_seed(seed)
    det = Detect(nc=nc, anchors=synth.DEFAULT_ANCHORS, ch=ch)
    det.stride = torch.tensor(synth.DEFAULT_STRIDES)
    det.anchors /= det.stride.view(-1, 1, 1)
    return det.to(dev).to(dtype).eval()


@pytest.mark.parametrize("nc,sizes,bs", [(3, (16, 8, 4), 2), (16, (40, 20, 10), 3), (15, (13, 7, 5), 1), (18, (128, 64, 32), 2)])
def test_detect_inference_matches_oracle_fp32(dev, nc, sizes, bs):
    ch = (8, 16, 32)
    det = make_detect(nc, ch, dev)
    feats = [torch.randn(bs, c, s, s + (i % 2)) for i, (c, s) in enumerate(zip(ch, sizes))]     # non-square maps too
    with torch.no_grad():
        z, xs = det([f.to(dev) for f in feats])
        convs = [det.m[i](feats[i].to(dev)) for i in range(3)]
    raw = [c.view(bs, det.na, det.no, c.shape[2], c.shape[3]).permute(0, 1, 3, 4, 2).contiguous() for c in convs]
    for a, b in zip(xs, raw):
        assert torch.equal(a, b)                                  # the permuted raw head is bit-exact
    ref = pyref.detect_decode([r.cpu() for r in raw], det.anchors.cpu(), det.stride)
    assert z.shape == ref.shape
    assert torch.allclose(z.cpu(), ref, rtol=2e-6, atol=1e-6), (z.cpu() - ref).abs().max()


def test_detect_golden_fixture_from_reference_module(dev):
    raw = [torch.from_numpy(G[f"detect_raw{i}"]) for i in range(3)]
    from yolov5_obb_amd import _lib
    import ctypes as C
    bs, na, _, _, no = raw[0].shape
    a_total = sum(r.shape[1] * r.shape[2] * r.shape[3] for r in raw)
    z = torch.empty((bs, a_total, no), device=dev)
    off = 0
    for i, r in enumerate(raw):
        conv = r.permute(0, 1, 4, 2, 3).contiguous().view(bs, na * no, r.shape[2], r.shape[3]).to(dev)   # back to the conv layout
        px = (synth.grid_anchors()[i] * synth.DEFAULT_STRIDES[i]).reshape(-1).tolist()
        arr = (C.c_float * len(px))(*px)
        xp = torch.empty(r.shape, device=dev)
        rc = _lib.lib().obb_detect_decode(_lib.ptr(conv), 0, bs, na, no, r.shape[2], r.shape[3], C.cast(arr, C.c_void_p),
                                          synth.DEFAULT_STRIDES[i], _lib.ptr(xp), _lib.ptr(z), a_total, off, _lib.stream_ptr(dev))
        assert rc == 0
        assert torch.equal(xp.cpu(), r)
        off += na * r.shape[2] * r.shape[3]
    assert np.allclose(z.cpu().numpy(), G["detect_z"], rtol=2e-6, atol=1e-6)


def test_detect_inference_fp16(dev):
    det = make_detect(16, (8, 16, 32), dev, torch.float16)
    feats = [torch.randn(2, c, s, s).half() for c, s in zip((8, 16, 32), (32, 16, 8))]
    with torch.no_grad():
        z, xs = det([f.to(dev) for f in feats])
        convs = [det.m[i](feats[i].to(dev)) for i in range(3)]
        raw = [c.view(2, det.na, det.no, c.shape[2], c.shape[3]).permute(0, 1, 3, 4, 2).contiguous() for c in convs]
        # the reference's own op sequence on the GPU in fp16 (models/yolo.py:71-79)
        zs = []
        for i, r in enumerate(raw):
            grid, ag = det._make_grid(r.shape[3], r.shape[2], i)
            y = r.sigmoid()
            y[..., 0:2] = (y[..., 0:2] * 2 - 0.5 + grid) * det.stride[i]
            y[..., 2:4] = (y[..., 2:4] * 2) ** 2 * ag
            zs.append(y.view(2, -1, det.no))
        ref = torch.cat(zs, 1)
    for a, b in zip(xs, raw):
        assert torch.equal(a, b)
    assert z.dtype == torch.float16
    d = (z.float() - ref.float()).abs()
    ulp = torch.maximum(ref.float().abs() * 2 ** -10, torch.tensor(2.0 ** -24, device=dev))
    assert (d <= ulp * 1.01).all(), (d / ulp).max()
    assert (d == 0).float().mean() > 0.98


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
@pytest.mark.parametrize("nc,sizes,bs", [(16, (40, 20, 10), 3), (15, (13, 7, 5), 1), (15, (128, 64, 32), 2)])
def test_detect_all_levels_in_one_launch_same_bytes(dev, dtype, nc, sizes, bs):
    """obb_detect_decode_levels (the default of Detect.forward) writes the same z, objectness column and permuted heads as one
    obb_detect_decode_col call per level -- vector and element-wise paths (13 x 14 maps are not 16-byte aligned), both dtypes."""
    ch = (8, 16, 32)
    det = make_detect(nc, ch, dev, dtype)
    feats = [torch.randn(bs, c, s, s + (i % 2)).to(dtype).to(dev) for i, (c, s) in enumerate(zip(ch, sizes))]
    outs = []
    with torch.no_grad():
        for fused in (True, False):
            det.fused_levels = fused
            z, xs = det(list(feats))
            outs.append((z, z._obb_objcol[0], xs))
    (z1, c1, x1), (z0, c0, x0) = outs
    assert torch.equal(z1, z0) and torch.equal(c1, c0) and torch.equal(c1, z1[..., 4])
    assert all(torch.equal(a, b) for a, b in zip(x1, x0))


def test_detect_rejects_cpu_inference_but_trains_on_cpu():
    from yolov5_obb_amd.models.yolo import Detect
    det = Detect(nc=3, anchors=synth.DEFAULT_ANCHORS, ch=(4, 4, 4))
    det.stride = torch.tensor(synth.DEFAULT_STRIDES)
    out = det([torch.zeros(1, 4, 8, 8), torch.zeros(1, 4, 4, 4), torch.zeros(1, 4, 2, 2)])     # training branch: plain torch
    assert [tuple(o.shape) for o in out] == [(1, 3, 8, 8, 188), (1, 3, 4, 4, 188), (1, 3, 2, 2, 188)]
    det.eval()
    with pytest.raises(RuntimeError):
        det([torch.zeros(1, 4, 8, 8), torch.zeros(1, 4, 4, 4), torch.zeros(1, 4, 2, 2)])


def test_csl_encode_matches_numpy_reference(dev):
    from yolov5_obb_amd.utils.rboxs_utils import csl_encode, gaussian_label_cpu
    ang = np.concatenate([G["csl_angles"], np.random.default_rng(0).uniform(0, 180, 500), [179.999, 0.0, 90.0, 180.0, 200.0, -5.0, 400.0]])
    for sig in (2.0, 4.0, 6.0):
        got = csl_encode(torch.from_numpy(ang.astype(np.float32)).to(dev), 180, 0.0, sig).cpu().numpy()
        ref = np.stack([pyref.gaussian_label(float(np.float32(a)), 180, 0, sig) for a in ang]).astype(np.float32)
        assert got.shape == ref.shape
        assert np.allclose(got, ref, rtol=1e-6, atol=1e-30), np.abs(got - ref).max()
        assert np.array_equal(got.argmax(1), ref.argmax(1))
        mine = np.stack([gaussian_label_cpu(float(np.float32(a)), 180, 0, sig) for a in ang]).astype(np.float32)
        assert np.array_equal(mine, ref)
    g37 = csl_encode(torch.from_numpy(G["csl_angles"].astype(np.float32)).to(dev), 180, 0.0, 2.0).cpu().numpy()
    ref37 = np.stack([pyref.gaussian_label(float(np.float32(a)), 180, 0, 2.0) for a in G["csl_angles"]])
    assert np.allclose(g37, ref37.astype(np.float32), rtol=1e-6, atol=1e-30)


def test_rbox2poly_and_hbb_match_reference(dev):
    from yolov5_obb_amd.utils.rboxs_utils import rbox2poly, poly2hbb, rbox2hbb
    rb = synth.s_uniform(300, 21)[0]
    poly = rbox2poly(rb.to(dev))
    assert np.allclose(poly.cpu().numpy(), G["rbox2poly_ref"], rtol=1e-6, atol=1e-4)
    hbb = rbox2hbb(rb.to(dev))
    assert np.allclose(hbb.cpu().numpy(), G["poly2hbb_ref"], rtol=1e-6, atol=1e-4)
    assert torch.allclose(poly2hbb(poly), hbb, rtol=0, atol=0)
    big = synth.s_clustered(100000, 300, 1)[0]
    p2 = rbox2poly(big.to(dev)).cpu()
    assert torch.allclose(p2, pyref.rbox2poly(big), rtol=1e-6, atol=2e-4)
    assert rbox2poly(torch.zeros(0, 5, device=dev)).shape == (0, 8)
