"""``obb_nms`` / ``poly_nms`` with the reference's signatures
(utils/nms_rotated/nms_rotated_wrapper.py:6-67), backed by libobb_hip.so.

Differences that are deliberate:
  * the too-small-box filter (``min(w,h) < 0.001``, wrapper:32-39) runs inside the
    kernel (OBB_NMS_DROP_SMALL) instead of three boolean-mask gathers and a host
    ``.all()`` sync; the returned indices are already indices into the input;
  * the returned ``inds`` lives on the device of ``dets`` (the reference's CPU
    ``ori_inds[cuda_inds]`` remap, wrapper:36-42, raises on torch >= 2).
"""
import numpy as np
import torch

from ... import _lib
from ... import nms_rotated_ext


def _to_tensor(dets, device_id):
    if isinstance(dets, torch.Tensor):
        return False, dets
    if isinstance(dets, np.ndarray):
        device = 'cpu' if device_id is None else f'cuda:{device_id}'
        return True, torch.from_numpy(dets).to(device)
    raise TypeError('dets must be eithr a Tensor or numpy array, '
                    f'but got {type(dets)}')


def obb_nms(dets, scores, iou_thr, device_id=None):
    """RIoU NMS (nms_rotated_wrapper.py:6-46).

    Args:
        dets (tensor/array): (num, [cx cy w h theta]) theta in radians, [-pi/2, pi/2)
        scores (tensor/array): (num)
        iou_thr (float)
    Returns:
        dets[inds, :], inds  -- inds in descending-score order
    """
    is_numpy, dets_th = _to_tensor(dets, device_id)
    if is_numpy:
        scores = torch.from_numpy(np.asarray(scores)).to(dets_th.device)
    if dets_th.numel() == 0:
        inds = dets_th.new_zeros(0, dtype=torch.int64)
    else:
        _lib.require_cuda(dets_th, "dets")
        if dets_th.dtype == torch.float64:      # the reference hands the tensors on as they are: double kernel (nms_rotated_cuda.cu:96)
            inds = nms_rotated_ext._run_rotated_f64(dets_th, scores.double(), iou_thr, flags=_lib.OBB_NMS_DROP_SMALL)
        else:
            d = dets_th.float() if dets_th.dtype != torch.float32 else dets_th
            s = scores.float() if scores.dtype != torch.float32 else scores
            inds = nms_rotated_ext._run_rotated(d.contiguous(), s.contiguous(), iou_thr, flags=_lib.OBB_NMS_DROP_SMALL)
    if is_numpy:
        inds = inds.cpu().numpy()
    return dets[inds, :], inds


def poly_nms(dets, iou_thr, device_id=None):
    """Polygon NMS (nms_rotated_wrapper.py:49-67). dets (num, [x1 y1 x2 y2 x3 y3 x4 y4 score])."""
    is_numpy, dets_th = _to_tensor(dets, device_id)
    if dets_th.device == torch.device('cpu'):
        raise NotImplementedError
    inds = nms_rotated_ext.nms_poly(dets_th.float(), iou_thr)
    if is_numpy:
        inds = inds.cpu().numpy()
    else:
        inds = inds.to(dets_th.device)
    return dets[inds, :], inds
