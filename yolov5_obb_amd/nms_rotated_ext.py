"""Mirror of the reference's compiled module ``nms_rotated_ext``
(utils/nms_rotated/src/nms_rotated_ext.cpp:57-60): ``nms_rotated`` and ``nms_poly``.

Same signatures, return conventions and error types; the work is done by
``obb_nms_rotated_f32`` / ``obb_nms_poly_f32`` of libobb_hip.so.  Two bindings of that C ABI exist: the compiled torch
extension ``nms_rotated_ext_c`` (csrc/torch_ext/nms_rotated_ext.cpp, pybind11 like the reference's module; used when built)
and the ctypes calls below (the fallback binding, ``OBB_BINDING=ctypes``).
"""
import torch

from . import _lib

__all__ = ["nms_rotated", "nms_poly"]
_ws_bytes = {}      # n -> obb_nms_workspace_bytes(n, 1, 0)


def _run_rotated(dets, scores, iou_threshold, flags=0, max_keep=0):
    ext = _lib.compiled()
    if ext is not None:          # the compiled binding: same entry point, checks and abort retry (csrc/torch_ext/nms_rotated_ext.cpp)
        return ext.nms_rotated_opts(dets, scores, float(iou_threshold), int(flags), int(max_keep))
    return _lib.retry_on_abort(lambda: _run_rotated_once(dets, scores, iou_threshold, flags, max_keep))


def _run_rotated_once(dets, scores, iou_threshold, flags, max_keep):
    L = _lib.lib()
    n = dets.shape[0]
    dev = dets.device
    keep = torch.empty(n, dtype=torch.int64, device=dev)
    cnt, cnt_view = _lib.pinned_count(dev)
    with _lib.guard(dev):
        st = _lib.stream_handle(dev)
        nbytes = _ws_bytes.get(n)
        if nbytes is None:
            if len(_ws_bytes) > 4096:
                _ws_bytes.clear()
            nbytes = _ws_bytes[n] = L.obb_nms_workspace_bytes(n, 1, 0)
        ws = _lib.workspace(nbytes, dev, st)
        rc = L.obb_nms_rotated_f32(_lib.ptr(dets), _lib.ptr(scores), n, float(iou_threshold), int(flags), int(max_keep),
                                   _lib.ptr(keep), _lib.ptr(cnt), _lib.ptr(ws), ws.numel(), _lib.C.c_void_p(st))
    _lib.check(rc, "obb_nms_rotated_f32")
    return keep[: _lib.checked_count(_lib.wait_count(cnt_view, dev), "obb_nms_rotated_f32")]


def nms_rotated(dets, scores, iou_threshold):
    """nms_rotated_ext.cpp:25-39.  dets (N,5) [cx,cy,w,h,theta_rad], scores (N) -> LongTensor of kept
    original indices in descending-score order, on dets' device.  A box is dropped iff an earlier kept box
    has IoU > iou_threshold (the CUDA path's strict compare, nms_rotated_cuda.cu:60)."""
    _lib.require_cuda(dets, "dets")
    _lib.require_cuda(scores, "scores")
    if dets.device != scores.device:
        raise RuntimeError("dets and scores must be on the same device")  # bare assert in the reference (:29)
    if dets.dtype != scores.dtype:
        raise RuntimeError("dets should have the same type as scores")  # nms_rotated_cpu.cpp:19-21
    if dets.dtype not in (torch.float32, torch.float64):
        # AT_DISPATCH_FLOATING_TYPES (nms_rotated_cuda.cu:96): float and double, no half
        raise RuntimeError(f"nms_rotated: float32 or float64 expected, got {dets.dtype}")
    if dets.dim() != 2 or dets.shape[1] != 5 or scores.dim() != 1 or scores.shape[0] != dets.shape[0]:
        raise RuntimeError(f"nms_rotated: expected dets (N,5) and scores (N), got {tuple(dets.shape)} {tuple(scores.shape)}")
    if dets.numel() == 0:
        return torch.empty(0, dtype=torch.int64, device=dets.device)
    if dets.dtype == torch.float64:
        return _run_rotated_f64(dets, scores, iou_threshold)
    return _run_rotated(dets.contiguous(), scores.contiguous(), iou_threshold)


def _run_rotated_f64(dets, scores, iou_threshold, flags=0, max_keep=0):
    """float64 input: the reference dispatches double to a double-precision instantiation of its kernel
    (nms_rotated_cuda.cu:96).  ``obb_nms_rotated_f64`` does the same on the device: double scores decide the order (ties:
    ascending index, NaN first), every IoU is evaluated in double (csrc/riou64_device.h, the policy RotGeom64 of the
    persistent NMS kernel) and compared with the float threshold of the kernel's signature (nms_rotated_cuda.cu:14,60)."""
    ext = _lib.compiled()
    if ext is not None:
        return ext.nms_rotated_opts(dets, scores, float(iou_threshold), int(flags), int(max_keep))
    return _lib.retry_on_abort(lambda: _run_rotated_f64_once(dets.contiguous(), scores.contiguous(), iou_threshold, flags, max_keep))


def _run_rotated_f64_once(dets, scores, iou_threshold, flags, max_keep):
    L = _lib.lib()
    n = dets.shape[0]
    dev = dets.device
    keep = torch.empty(n, dtype=torch.int64, device=dev)
    cnt, cnt_view = _lib.pinned_count(dev)
    with _lib.guard(dev):
        st = _lib.stream_handle(dev)
        ws = _lib.workspace(L.obb_nms_workspace_bytes(n, 1, 3), dev, st)
        rc = L.obb_nms_rotated_f64(_lib.ptr(dets), _lib.ptr(scores), n, float(iou_threshold), int(flags), int(max_keep),
                                   _lib.ptr(keep), _lib.ptr(cnt), _lib.ptr(ws), ws.numel(), _lib.C.c_void_p(st))
    _lib.check(rc, "obb_nms_rotated_f64")
    return keep[: _lib.checked_count(_lib.wait_count(cnt_view, dev), "obb_nms_rotated_f64")]


def nms_poly(dets, iou_threshold):
    """nms_rotated_ext.cpp:42-55.  dets (N,9) [x1 y1 .. x4 y4 score] float32 on the GPU -> LongTensor of kept
    original indices in descending-score order.  CPU input raises like the reference (AT_ERROR, :54)."""
    if not isinstance(dets, torch.Tensor):
        raise TypeError(f"dets must be a torch.Tensor, got {type(dets)}")
    if not dets.is_cuda:
        raise RuntimeError("POLY_NMS is not implemented on CPU")
    if dets.numel() == 0:
        # the reference returns a CPU tensor here (:47-48)
        return torch.empty(0, dtype=torch.int64, device="cpu")
    if dets.dtype != torch.float32:
        raise RuntimeError(f"nms_poly: float32 expected (the reference wrapper casts with .float()), got {dets.dtype}")
    if dets.dim() != 2 or dets.shape[1] < 9:
        raise RuntimeError(f"nms_poly: expected dets (N,9), got {tuple(dets.shape)}")
    ext = _lib.compiled()
    if ext is not None:
        return ext.nms_poly(dets, float(iou_threshold))
    dets = dets.contiguous()
    L = _lib.lib()
    n, stride = dets.shape[0], dets.shape[1]
    dev = dets.device
    keep = torch.empty(n, dtype=torch.int64, device=dev)
    cnt, cnt_view = _lib.pinned_count(dev)
    with _lib.guard(dev):
        ws = _lib.workspace(L.obb_nms_workspace_bytes(n, 1, 1), dev)
        rc = L.obb_nms_poly_f32(_lib.ptr(dets), stride, n, float(iou_threshold), 0, _lib.ptr(keep), _lib.ptr(cnt),
                                _lib.ptr(ws), ws.numel(), _lib.stream_ptr(dev))
    _lib.check(rc, "obb_nms_poly_f32")
    return keep[: _lib.checked_count(_lib.wait_count(cnt_view, dev), "obb_nms_poly_f32")]
