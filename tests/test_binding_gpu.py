"""GPU: the two bindings of the C ABI -- the compiled torch extension nms_rotated_ext_c (csrc/torch_ext/nms_rotated_ext.cpp; the
reference's nms_rotated_ext is a pybind11 torch extension, utils/nms_rotated/src/nms_rotated_ext.cpp:57-60) and the ctypes
fallback -- give the same results, and those are the oracle's.  Every other GPU test runs through whichever binding is active
(the compiled one when it is built); this file runs both side by side."""
import numpy as np
import pytest
import torch

from tests import synth

pytestmark = pytest.mark.gpu


@pytest.fixture
def ctypes_binding(monkeypatch):
    """Force the fallback binding for the duration of a test."""
    from yolov5_obb_amd import _lib
    _lib.compiled()
    monkeypatch.setattr(_lib, "_ext", None)
    monkeypatch.setattr(_lib, "_ext_tried", True)
    return _lib


def _active():
    from yolov5_obb_amd import _lib
    return _lib.compiled()


def test_compiled_binding_is_the_active_one(dev):
    ext = _active()
    assert ext is not None, "nms_rotated_ext_c.so is not built: the GPU suite would silently run on the ctypes fallback"
    from yolov5_obb_amd import _lib
    assert ext.library() == _lib.LIB_PATH


def test_nms_rotated_both_bindings_match_the_oracle(dev, oracle_lib, ctypes_binding):
    import oracle
    from yolov5_obb_amd import _lib, nms_rotated_ext
    d, s = synth.s_clustered(20000, 300, seed=3)
    s = synth.tie_free(s)
    ref = oracle.nms_rotated(d.numpy(), s.numpy(), 0.4)
    got_ct = nms_rotated_ext.nms_rotated(d.to(dev), s.to(dev), 0.4).cpu().numpy()
    assert np.array_equal(got_ct, ref)
    _lib._ext_tried = False                      # back to the compiled binding inside the same test
    assert _lib.compiled() is not None
    got_c = nms_rotated_ext.nms_rotated(d.to(dev), s.to(dev), 0.4).cpu().numpy()
    assert np.array_equal(got_c, ref)
    # float64 and the wrapper's small-box flag go through the same entry
    got64 = nms_rotated_ext.nms_rotated(d.double().to(dev), s.double().to(dev), 0.4).cpu().numpy()
    ref64 = oracle.nms_rotated_f64(d.double().numpy(), s.double().numpy(), 0.4) if hasattr(oracle, "nms_rotated_f64") else None
    if ref64 is not None:
        assert np.array_equal(got64, ref64)
    q9 = torch.cat((synth.rbox_to_quad(d[:5000]), s[:5000, None]), 1).contiguous()
    refq = oracle.nms_poly(q9.numpy(), 0.3) if hasattr(oracle, "nms_poly") else None
    gq = nms_rotated_ext.nms_poly(q9.to(dev), 0.3).cpu().numpy()
    if refq is not None:
        assert np.array_equal(gq, refq)
    assert nms_rotated_ext.nms_poly(torch.zeros(0, 9, device=dev), 0.3).device.type == "cpu"      # nms_rotated_ext.cpp:47-48


def test_fused_driver_both_bindings_match_the_oracle(dev, ctypes_binding):
    from oracle import pyref
    from yolov5_obb_amd import _lib
    from yolov5_obb_amd.utils.general import non_max_suppression_obb
    pred = synth.s_pred(3, 9000, 15, seed=11)
    kw = dict(conf_thres=0.25, iou_thres=0.45, multi_label=True, max_det=300)
    want = pyref.non_max_suppression_obb(pred.clone(), **kw)
    pd = pred.to(dev)
    for binding in ("ctypes", "compiled"):
        if binding == "compiled":
            _lib._ext_tried = False
            assert _lib.compiled() is not None
        for _ in range(3):                       # first call (no hint), hinted calls
            got = non_max_suppression_obb(pd, **kw)
            assert len(got) == len(want)
            for g, w in zip(got, want):
                assert np.array_equal(synth.canon_rows(g.cpu()), synth.canon_rows(w)), binding
    # classes filter, best-class mode, labels: the rarely used arguments through the compiled entry
    for kw2 in (dict(conf_thres=0.3, iou_thres=0.4, multi_label=False, max_det=100), dict(conf_thres=0.25, iou_thres=0.45, multi_label=True, classes=[1, 3, 7]),
                dict(conf_thres=0.25, iou_thres=0.45, agnostic=True, multi_label=True)):
        want2 = pyref.non_max_suppression_obb(pred.clone(), **kw2)
        got2 = non_max_suppression_obb(pd, **kw2)
        for g, w in zip(got2, want2):
            assert np.array_equal(synth.canon_rows(g.cpu()), synth.canon_rows(w)), kw2


def test_hint_hysteresis_keeps_results_exact_when_batches_hover_around_a_limit(dev):
    """Batches whose largest class segment alternates around the small-segment kernel's limit (384): every call is exact; after the
    first repeat the persistent kernel is held (no 2x on every other batch)."""
    from oracle import pyref
    from yolov5_obb_amd.utils.general import non_max_suppression_obb
    kw = dict(conf_thres=0.25, iou_thres=0.45, multi_label=True, max_det=1500)
    small = synth.s_pred(2, 20000, 4, seed=21, n_obj=40, fg_frac=0.03)
    big = synth.s_pred(2, 20000, 4, seed=22, n_obj=40, fg_frac=0.12)           # ~4x the candidates per class
    w_small = pyref.non_max_suppression_obb(small.clone(), **kw)
    w_big = pyref.non_max_suppression_obb(big.clone(), **kw)
    sd, bd = small.to(dev), big.to(dev)
    for i in range(6):
        for p, w in ((sd, w_small), (bd, w_big)):
            got = non_max_suppression_obb(p, **kw)
            for g, ww in zip(got, w):
                assert np.array_equal(synth.canon_rows(g.cpu()), synth.canon_rows(ww)), i


def test_val_tail_both_bindings_identical(dev, ctypes_binding):
    from yolov5_obb_amd import _lib
    from yolov5_obb_amd import val as V
    from yolov5_obb_amd.utils.general import non_max_suppression_obb
    pred = synth.s_pred(5, 9000, 15, seed=5).to(dev)
    _lib._ext_tried = False
    _lib.compiled()
    dets = non_max_suppression_obb(pred, conf_thres=0.25, iou_thres=0.45, multi_label=True, max_det=200)
    g = torch.Generator().manual_seed(1)
    tg = []
    for b in range(5):
        k = 7 if b != 2 else 0                                              # one image without labels
        if k:
            tg.append(torch.cat((torch.full((k, 1), float(b)), torch.randint(0, 15, (k, 1), generator=g).float(), torch.rand(k, 2, generator=g) * 1024,
                                 torch.rand(k, 1, generator=g) * 100 + 20, torch.rand(k, 1, generator=g) * 20 + 8, (torch.rand(k, 1, generator=g) - 0.5) * 3.14), 1))
    targets = torch.cat(tg, 0).to(dev)
    shapes = [((1024, 1024), ((0.9, 0.9), (3.0, 5.0)))] * 5
    iouv = torch.linspace(0.5, 0.95, 10, device=dev)
    out_c, (boxes_c, offs_c) = V.val_tail_batch(dets, targets, shapes, iouv, want_boxes=True)
    _lib._ext, _lib._ext_tried = None, True                                   # the ctypes binding
    out_p, (boxes_p, offs_p) = V.val_tail_batch(dets, targets, shapes, iouv, want_boxes=True)
    assert list(offs_c) == list(offs_p) and len(out_c) == len(out_p) == 5
    for (c1, f1, p1), (c2, f2, p2) in zip(out_c, out_p):
        assert c1.dtype == torch.bool and torch.equal(c1, c2) and torch.equal(f1, f2) and torch.equal(p1, p2)
    for a, b in zip(boxes_c, boxes_p):
        assert torch.equal(a, b)


@pytest.mark.parametrize("bs,A,nc,max_det,fg,n_obj", [
    (2, 6000, 80, 50, 0.06, 150),      # many classes, rows beyond max_det dropped by rank
    (20, 4000, 16, 300, 0.03, 60),     # 320 segments: more workgroups than compute units
    (4, 2000, 3, 300, 0.004, 30),      # a handful of rows per image (several lanes share one entry's lists), one image without any
    (1, 12000, 200, 1500, 0.05, 300),  # 200 segments of one image
    (16, 12000, 16, 1500, 0.03, 120),  # the shape of the headline step, smaller
])
def test_output_stage_inside_the_nms_kernel_matches_the_separate_one_and_the_oracle(dev, ctypes_binding, bs, A, nc, max_det, fg, n_obj):
    """Round 5: with out_packed = 0 (the compiled binding) the small-segment NMS kernel writes the output rows itself
    (csrc/nmsobb_impl.h: SmallGather); with out_packed = 1 (the ctypes binding) k_gather_out does, in a launch of its own.  Same
    rows, the oracle's (utils/general.py:772-862), on the first (un-hinted) call and on the hinted ones."""
    from oracle import pyref
    from yolov5_obb_amd import _lib
    from yolov5_obb_amd.utils.general import non_max_suppression_obb
    pred = synth.s_pred(bs, A, nc, seed=100 + bs + nc, n_obj=n_obj, fg_frac=fg)
    if bs == 4:
        pred[1, :, 4] = 0.0
    kw = dict(conf_thres=0.25, iou_thres=0.45, multi_label=True, max_det=max_det)
    want = [synth.canon_rows(w) for w in pyref.non_max_suppression_obb(pred.clone(), **kw)]
    pd = pred.to(dev)
    rows = {}
    for binding in ("ctypes", "compiled"):
        if binding == "compiled":
            _lib._ext_tried = False
            assert _lib.compiled() is not None
        for call in range(3):
            got = non_max_suppression_obb(pd, **kw)
            assert len(got) == bs
            for b, (g, w) in enumerate(zip(got, want)):
                assert g.shape[0] <= max_det
                assert np.array_equal(synth.canon_rows(g.cpu()), w), (binding, call, b)
        rows[binding] = [g.cpu() for g in got]
    for a, b in zip(rows["ctypes"], rows["compiled"]):       # the ORDER of the rows as well (descending score: the reference's single pass)
        assert torch.equal(a, b)
    if bs > 1:
        d0 = got[0].data_ptr()
        assert all(g.data_ptr() == d0 + b * max_det * 28 for b, g in enumerate(got) if g.shape[0]), "image b's rows start at row b * max_det"


def test_caller_kept_counters_are_left_zeroed_by_every_path_and_change_no_row(dev):
    """obb_non_max_suppression_obb_st (include/obb_hip.h): the candidate counters live in a buffer of the caller's, zeroed once;
    each call leaves them zeroed (no reset launch).  Through the C ABI: the un-hinted call (generic sort + persistent kernel +
    k_gather_out), the hinted one (in-LDS sort + small-segment kernel with its own output stage, out_packed = 0) and the hinted
    packed one (small-segment kernel + k_gather_out) give the rows of obb_non_max_suppression_obb_col, and the buffer reads zero
    after each of them."""
    import ctypes as C
    from yolov5_obb_amd import _lib
    L = _lib.lib()
    bs, A, nc, max_det = 6, 9000, 15, 300
    pred = synth.s_pred(bs, A, nc, seed=77, n_obj=60, fg_frac=0.04).to(dev)
    no = pred.shape[2]
    cap = A * nc
    ws = torch.empty(L.obb_nms_obb_workspace_bytes(bs, cap, nc, 0), dtype=torch.uint8, device=dev)
    state = torch.zeros(L.obb_nms_obb_state_bytes(bs), dtype=torch.uint8, device=dev)
    assert state.numel() >= bs * 260
    null = C.c_void_p(0)

    def call(hint, packed, kept):
        out = torch.zeros((bs * max_det, 7), dtype=torch.float32, device=dev)
        meta = torch.zeros(bs + 2, dtype=torch.int64, device=dev)
        args = [_lib.ptr(pred), null, 0, bs, A, no, 0.25, 0.45, null, 0, 0, 1, max_det, 30000, 4096.0, null, 0, cap, hint, _lib.ptr(out), packed,
                _lib.ptr(meta), C.c_void_p(meta.data_ptr() + 8 * bs), _lib.ptr(ws), ws.numel()]
        with _lib.guard(dev):
            st = C.c_void_p(_lib.stream_handle(dev))
            if kept:
                rc = L.obb_non_max_suppression_obb_st(*args, _lib.ptr(state), state.numel(), st)
            else:
                rc = L.obb_non_max_suppression_obb_col(*args, st)
        assert rc == 0
        torch.cuda.synchronize()
        m = meta.tolist()
        assert m[bs] == 0 and min(m[:bs]) >= 0, m
        rows, off = [], 0
        for b in range(bs):
            first = off if packed else b * max_det
            rows.append(out[first:first + m[b]].cpu())
            off += m[b]
        return rows, m[bs + 1]

    ref, st1 = call(0, 1, False)
    assert sum(len(r) for r in ref) > 300
    hint = (st1 & 0xffffffff) | (((st1 >> 32) & 0x1fffffff) << 32)
    assert 0 < ((st1 >> 32) & 0x1fffffff) <= 384 or ((st1 >> 32) & 0x1fffffff) == 0
    ref_h, st2 = call(hint, 1, False)
    hint = (st2 & 0xffffffff) | (((st2 >> 32) & 0x1fffffff) << 32)
    assert 0 < (hint >> 32) <= 384, "the hinted call reports the largest class segment: the next one takes the small-segment kernel"
    for h, packed in ((0, 1), (0, 0), (hint, 0), (hint, 1), (hint, 0)):
        got, _ = call(h, packed, True)
        assert int(state.count_nonzero()) == 0, (h, packed)
        for g, r in zip(got, ref):
            assert torch.equal(g, r), (h, packed)
    for g, r in zip(ref_h, ref):
        assert torch.equal(g, r)
    # a misaligned or short buffer is refused before anything is launched
    with _lib.guard(dev):
        st = C.c_void_p(_lib.stream_handle(dev))
        out = torch.zeros((bs * max_det, 7), dtype=torch.float32, device=dev)
        meta = torch.zeros(bs + 2, dtype=torch.int64, device=dev)
        args = [_lib.ptr(pred), null, 0, bs, A, no, 0.25, 0.45, null, 0, 0, 1, max_det, 30000, 4096.0, null, 0, cap, 0, _lib.ptr(out), 0,
                _lib.ptr(meta), C.c_void_p(meta.data_ptr() + 8 * bs), _lib.ptr(ws), ws.numel()]
        assert L.obb_non_max_suppression_obb_st(*args, _lib.ptr(state), 16, st) != 0
        assert L.obb_non_max_suppression_obb_st(*args, null, state.numel(), st) != 0
