"""CPU: the host-side rules of the Detect -> NMS coupling (utils/general.py::_objectness_column) and the synthetic head
generator the bench uses for that chain.  No GPU call is made: the rule only looks at tensor identity, version and geometry."""
import torch

from oracle import pyref
from tests import synth
from yolov5_obb_amd.utils import general as G


def _tagged(bs=2, A=50, no=5 + 3 + 180, dtype=torch.float32):
    z = torch.rand(bs, A, no, dtype=dtype)
    col = z[..., 4].contiguous()
    z._obb_objcol = (col, z._version)
    return z, col


def test_column_is_trusted_only_for_the_untouched_tensor_object():
    z, col = _tagged()
    assert G._objectness_column(z, z) is col
    assert G._objectness_column(z.clone(), z.clone()) is None                 # a copy carries no column
    assert G._objectness_column(z, z.contiguous()) is col                     # (contiguous() of a contiguous tensor is the tensor)
    zt = z.transpose(0, 1)
    assert G._objectness_column(zt, zt.contiguous()) is None                  # another object, other geometry
    z2, _ = _tagged()
    z2[..., 4] *= 0.5                                                         # in-place write through a view bumps the version
    assert G._objectness_column(z2, z2) is None
    z3, _ = _tagged()
    z3.mul_(1.0)
    assert G._objectness_column(z3, z3) is None
    for bad in (torch.zeros(2, 49), torch.zeros(2, 50, dtype=torch.float16), torch.zeros(50, 2).t()):
        z4, _ = _tagged()
        z4._obb_objcol = (bad, z4._version)                                   # wrong shape / dtype / layout
        assert G._objectness_column(z4, z4) is None
    z5, _ = _tagged()
    z5._obb_objcol = ("not a tensor", z5._version)
    assert G._objectness_column(z5, z5) is None


def test_planted_heads_are_deterministic_and_decode_to_overlapping_detections():
    a = synth.s_head(2, 15, (32, 16, 8), seed=5, n_obj=12)
    b = synth.s_head(2, 15, (32, 16, 8), seed=5, n_obj=12)
    assert all(torch.equal(x, y) for x, y in zip(a, b))
    assert [tuple(x.shape) for x in a] == [(2, 3 * 200, 32, 32), (2, 3 * 200, 16, 16), (2, 3 * 200, 8, 8)]
    raw = [x.view(2, 3, 200, x.shape[2], x.shape[3]).permute(0, 1, 3, 4, 2).contiguous() for x in a]
    z = pyref.detect_decode(raw, synth.grid_anchors(), synth.DEFAULT_STRIDES)
    passing = int((z[..., 4] > 0.25).sum())
    assert 12 * 2 * 4 < passing < 12 * 2 * 18 + 40                            # an object fires on <= 18 (cell, anchor) pairs
    out = pyref.non_max_suppression_obb(z.clone(), conf_thres=0.25, iou_thres=0.45, multi_label=True, max_det=300)
    kept = sum(o.shape[0] for o in out)
    assert 12 <= kept < passing                                               # neighbours of an object suppress each other


def test_inference_tensors_carry_no_column_and_do_not_raise():
    """under torch.inference_mode() a tensor tracks no version counter (`._version` raises), so the trust rule
    cannot be checked: the column is neither attached (models/yolo.py) nor trusted (utils/general.py)."""
    with torch.inference_mode():
        z = torch.rand(2, 50, 5 + 3 + 180)
        assert torch.is_inference(z)
        col = z[..., 4].contiguous()
        z._obb_objcol = (col, 0)                       # as if somebody had attached one anyway
        assert G._objectness_column(z, z) is None      # ... it is not trusted, and nothing raises


def test_detect_host_tables_follow_in_place_anchor_updates():
    """autoanchor / check_anchor_order rewrite m.anchors in place after the first inference call; the host copy
    Detect's kernel launch reads must follow (keyed on version + storage like ComputeLoss._refresh_host_tables)."""
    from yolov5_obb_amd.models.yolo import Detect
    det = Detect(nc=3, anchors=synth.DEFAULT_ANCHORS, ch=(8, 8, 8))
    det.stride = torch.tensor(synth.DEFAULT_STRIDES)
    det.anchors /= det.stride.view(-1, 1, 1)
    px0, st0 = det._host_tables()
    assert [round(v) for v in px0[0]] == synth.DEFAULT_ANCHORS[0] and st0 == synth.DEFAULT_STRIDES
    px1, _ = det._host_tables()
    assert list(px1[2]) == list(px0[2])                # unchanged tensors: the cached copy
    det.anchors[:] = det.anchors.flip(0)               # check_anchor_order (utils/autoanchor.py:17-25) does exactly this
    px2, _ = det._host_tables()
    assert [round(v / 8.0 * 32.0) for v in px2[0]] == synth.DEFAULT_ANCHORS[2]
    import pickle
    det2 = pickle.loads(pickle.dumps(det))             # the cache holds plain lists + ints: still picklable
    assert [round(v) for v in det2._host_tables()[0][1]] == [round(v) for v in px2[1]]
