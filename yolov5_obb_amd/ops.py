"""Tensor-level access to the pairwise IoU entry points of libobb_hip.so.

``rotated_iou_*`` is the device function behind the NMS
(single_box_iou_rotated<float>, utils/nms_rotated/src/box_iou_rotated_utils.h:333-360);
``quad_iou_matrix`` is devPolyIoU (utils/nms_rotated/src/poly_nms_cuda.cu:122-142);
``rbox_overlaps`` is the devkit's overlaps_kernel on device tensors
(DOTA_devkit/poly_nms_gpu/poly_overlaps_kernel.cu:280-353).
"""
import torch

from . import _lib


def _prep(t, name, cols):
    _lib.require_cuda(t, name)
    if t.dim() != 2 or t.shape[1] < cols:
        raise RuntimeError(f"{name}: expected (N, >={cols}), got {tuple(t.shape)}")
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


def rotated_iou_pairs(a, b):
    a, b = _prep(a, "a", 5), _prep(b, "b", 5)
    if a.shape != b.shape or a.shape[1] != 5:
        raise RuntimeError("rotated_iou_pairs: a and b must both be (N,5)")
    out = torch.empty(a.shape[0], dtype=torch.float32, device=a.device)
    with torch.cuda.device(a.device):
        rc = _lib.lib().obb_rotated_iou_pairs_f32(_lib.ptr(a), _lib.ptr(b), a.shape[0], _lib.ptr(out), _lib.stream_ptr(a.device))
    _lib.check(rc, "obb_rotated_iou_pairs_f32")
    return out


def rotated_iou_matrix(a, b):
    a, b = _prep(a, "a", 5), _prep(b, "b", 5)
    if a.shape[1] != 5 or b.shape[1] != 5:
        raise RuntimeError("rotated_iou_matrix: (N,5) x (K,5) expected")
    out = torch.empty(a.shape[0], b.shape[0], dtype=torch.float32, device=a.device)
    with torch.cuda.device(a.device):
        rc = _lib.lib().obb_rotated_iou_matrix_f32(_lib.ptr(a), a.shape[0], _lib.ptr(b), b.shape[0], _lib.ptr(out),
                                                   _lib.stream_ptr(a.device))
    _lib.check(rc, "obb_rotated_iou_matrix_f32")
    return out


def quad_iou_matrix(a, b):
    a, b = _prep(a, "a", 8), _prep(b, "b", 8)
    out = torch.empty(a.shape[0], b.shape[0], dtype=torch.float32, device=a.device)
    with torch.cuda.device(a.device):
        rc = _lib.lib().obb_quad_iou_matrix_f32(_lib.ptr(a), a.shape[1], a.shape[0], _lib.ptr(b), b.shape[1], b.shape[0],
                                                _lib.ptr(out), _lib.stream_ptr(a.device))
    _lib.check(rc, "obb_quad_iou_matrix_f32")
    return out


def rbox_overlaps(boxes, query):
    boxes, query = _prep(boxes, "boxes", 5), _prep(query, "query", 5)
    out = torch.empty(boxes.shape[0], query.shape[0], dtype=torch.float32, device=boxes.device)
    with torch.cuda.device(boxes.device):
        rc = _lib.lib().obb_rbox_overlaps_f32(_lib.ptr(boxes), boxes.shape[0], _lib.ptr(query), query.shape[0], _lib.ptr(out),
                                              _lib.stream_ptr(boxes.device))
    _lib.check(rc, "obb_rbox_overlaps_f32")
    return out
