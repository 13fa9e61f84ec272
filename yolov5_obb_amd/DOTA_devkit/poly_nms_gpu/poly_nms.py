"""``poly_gpu_nms`` with the signature of the reference's Cython module
(DOTA_devkit/poly_nms_gpu/poly_nms.pyx:9-24), bound with ctypes to the ``_poly_nms``
symbol of libobb_hip.so (same name and argument order as poly_nms.hpp:9-10)."""
import ctypes as C

import numpy as np

from ... import _lib


def poly_gpu_nms(dets, thresh, device_id=0):
    """dets (N,9) float32 [x1 y1 .. x4 y4 score] on the host -> python list of kept original indices."""
    dets = np.ascontiguousarray(dets, dtype=np.float32)
    boxes_num, boxes_dim = dets.shape[0], dets.shape[1]
    keep = np.zeros(boxes_num, dtype=np.int32)
    num_out = C.c_int(0)
    scores = dets[:, 8]
    order = scores.argsort()[::-1]                 # poly_nms.pyx:18-19 (numpy's tie order, as the reference)
    sorted_dets = np.ascontiguousarray(dets[order, :])
    if boxes_num:
        _lib.lib()._poly_nms(keep.ctypes.data_as(C.c_void_p), C.byref(num_out), sorted_dets.ctypes.data_as(C.c_void_p),
                             boxes_num, boxes_dim, float(thresh), int(device_id))
    keep = keep[:num_out.value]
    return list(order[keep])
