"""GPU, end to end: the band of val.py between the last convolution and the metric code, in the reference's own call order
(models/yolo.py:61-79 -> utils/general.py:772-862 -> val.py:226-250), every stage from this package, against the same chain
built from the oracle.  The head logits are synthetic (objects planted in random maps) so that the NMS has real work.

Decoded coordinates differ by a few fp32 ulps between the device's and the host's libm (tests/test_head_gpu.py), which could
flip a box that sits exactly on a threshold; the seeded inputs here have no such box, and the comparison is exact on
counts, classes and the correct-matrix, 1e-4 px on coordinates."""
import numpy as np
import pytest
import torch

from oracle import pyref
from tests import synth
from tests.test_head_gpu import make_detect

pytestmark = pytest.mark.gpu


def _planted_head(bs, na, no, sizes, nc, seed):
    """Raw head outputs (bs, na, ny, nx, no) with a few confident cells per image."""
    g = torch.Generator().manual_seed(seed)
    xs = []
    for n in sizes:
        x = torch.randn(bs, na, n, n, no, generator=g) * 0.5
        x[..., 4] -= 6.0                                   # background objectness ~ 0.002
        k = max(2, n // 3)
        for b in range(bs):
            for _ in range(k):
                a, gy, gx = int(torch.randint(0, na, (1,), generator=g)), int(torch.randint(0, n, (1,), generator=g)), int(torch.randint(0, n, (1,), generator=g))
                cls, ang = int(torch.randint(0, nc, (1,), generator=g)), int(torch.randint(0, 180, (1,), generator=g))
                for dy, dx in ((0, 0), (0, 1), (1, 0)):     # neighbouring cells see the same object
                    yy, xx = min(gy + dy, n - 1), min(gx + dx, n - 1)
                    x[b, a, yy, xx, 4] = 3.0 + torch.rand(1, generator=g).item()
                    x[b, a, yy, xx, 5 + cls] = 2.5 + torch.rand(1, generator=g).item()
                    x[b, a, yy, xx, 5 + nc + ang] = 4.0
        xs.append(x)
    return xs


@pytest.mark.parametrize("nc,sizes,bs", [(15, (32, 16, 8), 3), (16, (64, 32, 16), 2)])
def test_detect_nms_valtail_chain_matches_the_oracle_chain(dev, oracle_lib, nc, sizes, bs):
    from yolov5_obb_amd import _lib
    from yolov5_obb_amd.utils.general import non_max_suppression_obb
    from yolov5_obb_amd.val import process_batch, val_postprocess
    import ctypes as C
    na, no = 3, 5 + nc + 180
    raw = _planted_head(bs, na, no, sizes, nc, seed=nc)
    anchors = synth.grid_anchors()
    # ---- reference chain on the CPU (oracle)
    z_ref = pyref.detect_decode(raw, anchors, synth.DEFAULT_STRIDES)
    kw = dict(conf_thres=0.25, iou_thres=0.45, multi_label=True, max_det=300)
    det_ref = pyref.non_max_suppression_obb(z_ref.clone(), **kw)
    # ---- this package on the GPU: conv layout in, decode kernel, fused NMS, val tail
    a_total = sum(na * n * n for n in sizes)
    z = torch.empty((bs, a_total, no), device=dev)
    off = 0
    for i, r in enumerate(raw):
        conv = r.permute(0, 1, 4, 2, 3).contiguous().view(bs, na * no, r.shape[2], r.shape[3]).to(dev)
        px = (anchors[i] * synth.DEFAULT_STRIDES[i]).reshape(-1).tolist()
        arr = (C.c_float * len(px))(*px)
        rc = _lib.lib().obb_detect_decode(_lib.ptr(conv), 0, bs, na, no, r.shape[2], r.shape[3], C.cast(arr, C.c_void_p),
                                          float(synth.DEFAULT_STRIDES[i]), None, _lib.ptr(z), a_total, off, _lib.stream_ptr(dev))
        assert rc == 0
        off += na * r.shape[2] * r.shape[3]
    det = non_max_suppression_obb(z, **kw)
    iouv = torch.linspace(0.5, 0.95, 10)
    n_total = 0
    for b in range(bs):
        d, dr = det[b].cpu(), det_ref[b]
        assert d.shape == dr.shape and d.shape[0] > 0, (b, d.shape, dr.shape)
        assert torch.equal(d[:, 6], dr[:, 6]) and torch.allclose(d, dr, rtol=2e-6, atol=1e-4)
        n_total += d.shape[0]
        # val.py:226-250: polygons / horizontal boxes in both coordinate systems, then the match against labels
        gain, pad = 0.8125, (8.0, 24.0)
        out = val_postprocess(det[b], ratio_pad=((gain, gain), pad))
        ref = pyref.val_postprocess(dr.clone(), gain, pad)
        for o, r_ in zip(out, ref):
            assert torch.allclose(o.cpu(), r_, rtol=1e-6, atol=2e-4)
        labels = torch.cat((ref[3][::2, 5:6], ref[3][::2, :4] + 1.5), 1)       # every second detection is a (shifted) label
        got = process_batch(out[3], labels.to(dev), iouv.to(dev))
        assert torch.equal(got.cpu(), pyref.process_batch(ref[3], labels, iouv))
    assert n_total >= 3 * bs


@pytest.mark.parametrize("dtype,nc", [(torch.float32, 15), (torch.float16, 16), (torch.float16, 15)])
def test_detect_hands_its_objectness_column_to_the_nms(dev, dtype, nc):
    """Detect stores z[..., 4] densely next to z; non_max_suppression_obb reads its confidence filter from that column
    when it gets the very tensor Detect returned, untouched -- and must return exactly what the plain path returns."""
    from yolov5_obb_amd.utils import general as G
    sizes, bs, ch = (32, 16, 8), 3, (8, 16, 32)
    det = make_detect(nc, ch, dev, dtype, seed=nc)
    raw = _planted_head(bs, det.na, det.no, sizes, nc, seed=7 + nc)
    # feed the conv outputs directly: Detect.m replaced by identities over pre-made conv-layout tensors
    convs = [r.permute(0, 1, 4, 2, 3).contiguous().view(bs, det.na * det.no, r.shape[2], r.shape[3]).to(dev).to(dtype) for r in raw]
    det.m = torch.nn.ModuleList([torch.nn.Identity() for _ in convs])
    with torch.no_grad():
        z, _ = det(list(convs))
    tag = getattr(z, "_obb_objcol", None)
    assert tag is not None and torch.equal(tag[0], z[..., 4]) and tag[0].is_contiguous()      # bit-identical column
    kw = dict(conf_thres=0.25, iou_thres=0.45, multi_label=True, max_det=300)
    assert G._objectness_column(z, z) is tag[0]
    coupled = G.non_max_suppression_obb(z, **kw)
    plain = G.non_max_suppression_obb(z.clone(), **kw)                                         # a clone carries no column
    assert G._objectness_column(z.clone(), z.clone()) is None
    assert sum(o.shape[0] for o in plain) >= 3 * bs
    for a, b in zip(coupled, plain):
        assert torch.equal(a, b)
    # a column that no longer describes the tensor must not be used: zero every objectness in place -> no detections
    z[..., 4] = 0
    assert G._objectness_column(z, z) is None
    assert all(o.shape[0] == 0 for o in G.non_max_suppression_obb(z, **kw))
    # and the C entry point with a column that passes NOTHING returns nothing (the column is what the filter reads)
    with torch.no_grad():
        z2, _ = det(list(convs))
    z2._obb_objcol = (torch.zeros_like(z2._obb_objcol[0]), z2._version)
    assert all(o.shape[0] == 0 for o in G.non_max_suppression_obb(z2, **kw))
    det.couple_nms = False
    with torch.no_grad():
        z3, _ = det(list(convs))
    assert not hasattr(z3, "_obb_objcol") and torch.equal(z3, z2)
