"""Mirror of DOTA_devkit/poly_nms_gpu/nms_wrapper.py:11-16."""
from .poly_nms import poly_gpu_nms


def poly_nms_gpu(dets, thresh, force_cpu=False):
    """Dispatch to the GPU polygon NMS; [] on empty input (nms_wrapper.py:14-15)."""
    if dets.shape[0] == 0:
        return []
    return poly_gpu_nms(dets, thresh, device_id=0)
