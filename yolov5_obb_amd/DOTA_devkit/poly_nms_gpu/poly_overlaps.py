"""``poly_overlaps`` with the signature of the reference's Cython module
(DOTA_devkit/poly_nms_gpu/poly_overlaps.pyx:7-12), bound with ctypes to the
``_overlaps`` symbol of libobb_hip.so (same name and argument order as
poly_overlaps.hpp:1)."""
import ctypes as C

import numpy as np

from ... import _lib


def poly_overlaps(boxes, query_boxes, device_id=0):
    """boxes (N,5) float32, query_boxes (K,5) float32 [cx,cy,w,h,theta_rad] on the host -> (N,K) float32 IoU."""
    boxes = np.ascontiguousarray(boxes, dtype=np.float32)
    query_boxes = np.ascontiguousarray(query_boxes, dtype=np.float32)
    if boxes.ndim != 2 or query_boxes.ndim != 2:
        raise ValueError("Buffer has wrong number of dimensions (expected 2)")   # Cython's buffer check
    n, k = boxes.shape[0], query_boxes.shape[0]
    overlaps = np.zeros((n, k), dtype=np.float32)
    if n and k:
        _lib.lib()._overlaps(overlaps.ctypes.data_as(C.c_void_p), boxes.ctypes.data_as(C.c_void_p),
                             query_boxes.ctypes.data_as(C.c_void_p), n, k, int(device_id))
    return overlaps
