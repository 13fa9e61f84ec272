"""Mirror of the per-image tail of the reference's ``val.py``: what runs for every image right
after ``non_max_suppression_obb`` -- ``rbox2poly`` -> ``poly2hbb`` -> ``xywh2xyxy`` -> ``scale_polys`` (val.py:226-236) and
``process_batch`` (val.py:69-90) -- as two fused launches of libobb_hip.so instead of ~15 small ATen kernels and a
``.cpu().numpy()`` round trip per image."""
import torch

from . import _lib


def val_postprocess(pred, ratio_pad=None, img1_shape=None, img0_shape=None):
    """pred (n,7) [x y l s theta conf cls] (CUDA) -> pred_poly (n,10), pred_hbb (n,6), pred_polyn (n,10), pred_hbbn (n,6).

    ratio_pad = ((h_ratio, w_ratio), (pad_w, pad_h)) as val.py passes it (shapes[si][1]); when None it is derived from the
    two shapes exactly like utils/general.py:639-641."""
    _lib.require_cuda(pred, "pred")
    if pred.dim() != 2 or pred.shape[1] != 7:
        raise RuntimeError(f"val_postprocess: pred must be (n,7), got {tuple(pred.shape)}")
    if ratio_pad is None:
        gain = min(img1_shape[0] / img0_shape[0], img1_shape[1] / img0_shape[1])
        pad = (img1_shape[1] - img0_shape[1] * gain) / 2, (img1_shape[0] - img0_shape[0] * gain) / 2
    else:
        gain, pad = ratio_pad[0][0], ratio_pad[1]
    p = pred.to(torch.float32).contiguous()
    n, dev = p.shape[0], p.device
    poly = torch.empty((n, 10), dtype=torch.float32, device=dev)
    hbb = torch.empty((n, 6), dtype=torch.float32, device=dev)
    polyn = torch.empty((n, 10), dtype=torch.float32, device=dev)
    hbbn = torch.empty((n, 6), dtype=torch.float32, device=dev)
    with _lib.guard(dev):
        rc = _lib.lib().obb_val_postprocess_f32(_lib.ptr(p), n, float(pad[0]), float(pad[1]), float(gain), _lib.ptr(poly), _lib.ptr(hbb),
                                                _lib.ptr(polyn), _lib.ptr(hbbn), _lib.stream_ptr(dev))
    _lib.check(rc, "obb_val_postprocess_f32")
    return poly, hbb, polyn, hbbn


def process_batch(detections, labels, iouv):
    """Return correct predictions matrix (val.py:69-90).  Both sets of boxes are in (x1, y1, x2, y2) format.
    Arguments:
        detections (Array[N, 6]), x1, y1, x2, y2, conf, class
        labels (Array[M, 5]), class, x1, y1, x2, y2
    Returns:
        correct (Array[N, 10]), for 10 IoU levels
    """
    _lib.require_cuda(detections, "detections")
    dev = detections.device
    det = detections.to(torch.float32).contiguous()
    lab = labels.to(device=dev, dtype=torch.float32).contiguous()
    iv = iouv.to(device=dev, dtype=torch.float32).contiguous()
    n, m, niou = det.shape[0], lab.shape[0], iv.shape[0]
    correct = torch.zeros((n, niou), dtype=torch.bool, device=dev)
    if n == 0:
        return correct
    L = _lib.lib()
    with _lib.guard(dev):
        ws = torch.empty(L.obb_process_batch_workspace_bytes(n, m), dtype=torch.uint8, device=dev)
        out = correct.view(torch.uint8)
        rc = L.obb_process_batch_f32(_lib.ptr(det), n, _lib.ptr(lab) if m else 0, m, _lib.ptr(iv), niou, _lib.ptr(out), _lib.ptr(ws),
                                     ws.numel(), _lib.stream_ptr(dev))
    _lib.check(rc, "obb_process_batch_f32")
    return correct
