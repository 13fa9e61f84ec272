"""Mirror of the reference's ``utils/rboxs_utils.py`` (Oriented Bounding Boxes utils).

``gaussian_label_cpu``, ``regular_theta``, ``rbox2poly``, ``poly2hbb`` and ``poly_filter`` keep the reference's
signatures and numpy/torch behaviour (they are small host/elementwise helpers used by the dataloader, val.py and
detect.py); the device-side CSL encode used by the loss path lives in ``csl_encode`` (libobb_hip.so).
``poly2rbox`` needs OpenCV's minAreaRect exactly like the reference (utils/rboxs_utils.py:61) and raises a clear
ImportError when cv2 is not installed.
"""
import numpy as np
import torch

from .. import _lib

pi = 3.141592  # utils/rboxs_utils.py:5


def csl_encode(angles, num_class=180, u=0.0, sig=4.0):
    """Device-side batch form of ``gaussian_label_cpu``: angles (n,) CUDA tensor in [0, num_class) -> (n, num_class)
    float32 Circular Smooth Labels, one ``obb_csl_encode_f32`` launch (what utils/datasets.py:639-642 computes per
    sample on the CPU through poly2rbox)."""
    _lib.require_cuda(angles, "angles")
    a = angles.reshape(-1).to(torch.float32).contiguous()
    out = torch.empty((a.shape[0], int(num_class)), dtype=torch.float32, device=a.device)
    with torch.cuda.device(a.device):
        rc = _lib.lib().obb_csl_encode_f32(_lib.ptr(a), a.shape[0], int(num_class), float(u), float(sig), _lib.ptr(out),
                                           _lib.stream_ptr(a.device))
    _lib.check(rc, "obb_csl_encode_f32")
    return out


def _rbox2poly_device(obboxes, want_poly=True, want_hbb=False):
    r = obboxes.to(torch.float32).contiguous()
    n = r.shape[0]
    poly = torch.empty((n, 8), dtype=torch.float32, device=r.device) if want_poly else None
    hbb = torch.empty((n, 4), dtype=torch.float32, device=r.device) if want_hbb else None
    with torch.cuda.device(r.device):
        rc = _lib.lib().obb_rbox2poly_f32(_lib.ptr(r), n, r.shape[1], _lib.ptr(poly), _lib.ptr(hbb), _lib.stream_ptr(r.device))
    _lib.check(rc, "obb_rbox2poly_f32")
    return poly, hbb


def rbox2hbb(obboxes):
    """rbox2poly followed by poly2hbb in one kernel for CUDA tensors (val.py:226-241 uses the pair back to back)."""
    _lib.require_cuda(obboxes, "obboxes")
    return _rbox2poly_device(obboxes, want_poly=False, want_hbb=True)[1]


def gaussian_label_cpu(label, num_class, u=0, sig=4.0):
    """Circular Smooth Label of one angle (utils/rboxs_utils.py:9-26): a Gaussian window over
    x = -num_class/2 .. num_class/2-1, rolled so that its peak sits on the angle's bin."""
    x = np.arange(-num_class / 2, num_class / 2)
    y_sig = np.exp(-(x - u) ** 2 / (2 * sig ** 2))
    index = int(num_class / 2 - label)
    return np.concatenate([y_sig[index:], y_sig[:index]], axis=0)


def regular_theta(theta, mode='180', start=-pi / 2):
    """limit theta ∈ [-pi/2, pi/2) (utils/rboxs_utils.py:28-37)"""
    assert mode in ['360', '180']
    cycle = 2 * pi if mode == '360' else pi
    theta = theta - start
    theta = theta % cycle
    return theta + start


def poly2rbox(polys, num_cls_thata=180, radius=6.0, use_pi=False, use_gaussian=False):
    """poly (n,8) -> long-edge rbox (n,[cx cy l s θ]) [+ CSL labels] (utils/rboxs_utils.py:39-81)."""
    try:
        import cv2
    except ImportError as e:  # the reference imports cv2 at module import time (utils/rboxs_utils.py:6)
        raise ImportError("poly2rbox needs OpenCV (cv2.minAreaRect), as in the reference") from e
    assert polys.shape[-1] == 8
    csl_labels, rboxes = [], []
    for poly in polys:
        poly = np.float32(poly.reshape(4, 2))
        (x, y), (w, h), angle = cv2.minAreaRect(poly)
        theta = -angle / 180 * pi
        if w != max(w, h):
            w, h = h, w
            theta += pi / 2
        theta = regular_theta(theta)
        angle = (theta * 180 / pi) + 90
        rboxes.append([x, y, w, h, theta if use_pi else angle])
        if use_gaussian:
            csl_labels.append(gaussian_label_cpu(label=angle, num_class=num_cls_thata, u=0, sig=radius))
    if use_gaussian:
        return np.array(rboxes), np.array(csl_labels)
    return np.array(rboxes)


def rbox2poly(obboxes):
    """(…,[cx cy l s θ]) θ∈[-pi/2, pi/2) -> (…,[x1 y1 x2 y2 x3 y3 x4 y4]) (utils/rboxs_utils.py:106-145)."""
    if isinstance(obboxes, torch.Tensor) and obboxes.is_cuda and obboxes.dim() == 2 and obboxes.dtype == torch.float32 \
            and not obboxes.requires_grad:
        return _rbox2poly_device(obboxes)[0]      # one fused kernel (csrc/head.hip)
    if isinstance(obboxes, torch.Tensor):
        center, w, h, theta = obboxes[:, :2], obboxes[:, 2:3], obboxes[:, 3:4], obboxes[:, 4:5]
        Cos, Sin = torch.cos(theta), torch.sin(theta)
        vector1 = torch.cat((w / 2 * Cos, -w / 2 * Sin), dim=-1)
        vector2 = torch.cat((-h / 2 * Sin, -h / 2 * Cos), dim=-1)
        pts = (center + vector1 + vector2, center + vector1 - vector2, center - vector1 - vector2, center - vector1 + vector2)
        return torch.cat(pts, dim=-1).reshape(*obboxes.shape[:-1], 8)
    center, w, h, theta = np.split(obboxes, (2, 3, 4), axis=-1)
    Cos, Sin = np.cos(theta), np.sin(theta)
    vector1 = np.concatenate([w / 2 * Cos, -w / 2 * Sin], axis=-1)
    vector2 = np.concatenate([-h / 2 * Sin, -h / 2 * Cos], axis=-1)
    pts = [center + vector1 + vector2, center + vector1 - vector2, center - vector1 - vector2, center - vector1 + vector2]
    return np.concatenate(pts, axis=-1).reshape(*obboxes.shape[:-1], 8)


def poly2hbb(polys):
    """(n,8) polygons -> (n,[xc yc w h]) axis-aligned hulls (utils/rboxs_utils.py:147-181)."""
    assert polys.shape[-1] == 8
    x, y = polys[:, 0::2], polys[:, 1::2]
    if isinstance(polys, torch.Tensor):
        x_max, x_min, y_max, y_min = torch.amax(x, dim=1), torch.amin(x, dim=1), torch.amax(y, dim=1), torch.amin(y, dim=1)
        x_ctr, y_ctr = (x_max + x_min) / 2.0, (y_max + y_min) / 2.0
        return torch.cat((x_ctr.reshape(-1, 1), y_ctr.reshape(-1, 1), (x_max - x_min).reshape(-1, 1), (y_max - y_min).reshape(-1, 1)), dim=1)
    x_max, x_min, y_max, y_min = np.amax(x, axis=1), np.amin(x, axis=1), np.amax(y, axis=1), np.amin(y, axis=1)
    x_ctr, y_ctr = (x_max + x_min) / 2.0, (y_max + y_min) / 2.0
    return np.concatenate((x_ctr.reshape(-1, 1), y_ctr.reshape(-1, 1), (x_max - x_min).reshape(-1, 1), (y_max - y_min).reshape(-1, 1)), axis=1)


def poly_filter(polys, h, w):
    """Keep polygons whose hull centre lies inside the image (utils/rboxs_utils.py:183-199)."""
    x, y = polys[:, 0::2], polys[:, 1::2]
    x_ctr = (np.amax(x, axis=1) + np.amin(x, axis=1)) / 2.0
    y_ctr = (np.amax(y, axis=1) + np.amin(y, axis=1)) / 2.0
    return (x_ctr > 0) & (x_ctr < w) & (y_ctr > 0) & (y_ctr < h)
