"""Mirror of the reference's ``utils/rboxs_utils.py`` (Oriented Bounding Boxes utils).

``gaussian_label_cpu``, ``regular_theta``, ``rbox2poly``, ``poly2hbb`` and ``poly_filter`` keep the reference's
signatures and numpy/torch behaviour (they are small host/elementwise helpers used by the dataloader, val.py and
detect.py); the device-side CSL encode used by the loss path lives in ``csl_encode`` (libobb_hip.so).
``poly2rbox`` calls OpenCV's minAreaRect like the reference (utils/rboxs_utils.py:61) when cv2 is installed; without it
(this image) the minimum-area rectangle comes from ``_min_area_rect`` below -- the same definition, rotating calipers over the
hull's edges in double precision; the long-edge normalisation that follows does not depend on which of the equivalent
(w, h, angle) descriptions of a rectangle it is handed.
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


_warned_no_cv2 = False


def _min_area_rect(pts):
    """Minimum-area enclosing rectangle of a few points, ((cx, cy), (w, h), angle in degrees) in the convention of
    cv2.minAreaRect since OpenCV 4.5.1: w is the extent along the direction `angle`, measured from the x axis towards the
    y axis of the image frame, angle in (0, 90].  One of the rectangle's sides lies on an edge of the convex hull."""
    p = np.asarray(pts, dtype=np.float64).reshape(-1, 2)
    q = sorted(set(map(tuple, p.tolist())))
    if len(q) == 1:
        return (q[0][0], q[0][1]), (0.0, 0.0), 90.0

    def half(seq):
        h = []
        for v in seq:
            while len(h) >= 2 and (h[-1][0] - h[-2][0]) * (v[1] - h[-2][1]) - (h[-1][1] - h[-2][1]) * (v[0] - h[-2][0]) <= 0:
                h.pop()
            h.append(v)
        return h[:-1]
    hull = np.array(half(q) + half(q[::-1]))                     # monotone chain; two points for a degenerate set
    def normalised(w, h, ang):                                   # angle into (0, 90], sides swapped with every quarter turn
        k = int(np.floor(ang / 90.0))
        ang -= 90.0 * k
        if k % 2:
            w, h = h, w
        if ang <= 0.0:                                          # (0, 90]: an axis-aligned rectangle reports 90
            ang += 90.0
            w, h = h, w
        return w, h, ang
    cands = []
    for i in range(len(hull)):
        d = hull[(i + 1) % len(hull)] - hull[i]
        n = float(np.hypot(d[0], d[1]))
        if n == 0.0:
            continue
        u = d / n
        v = np.array([-u[1], u[0]])
        a, b = hull @ u, hull @ v
        w, h = float(a.max() - a.min()), float(b.max() - b.min())
        c = u * (a.max() + a.min()) / 2 + v * (b.max() + b.min()) / 2
        cands.append((w * h, c) + normalised(w, h, float(np.degrees(np.arctan2(u[1], u[0])))))
    # Tie rule (this function's own; OpenCV's rotating calipers work in float32 and keep whichever edge they meet first): among
    # the hull edges whose rectangle is within 1e-12 of the smallest area, the one with the smallest normalised angle.  The four
    # edges of a rectangle give the same normalised (w, h, angle) up to rounding, so the rule only decides genuinely different
    # rectangles of equal area.
    amin = min(t[0] for t in cands)
    _, c, w, h, ang = min((t for t in cands if t[0] <= amin * (1.0 + 1e-12) + 1e-300), key=lambda t: t[4])
    return (float(c[0]), float(c[1])), (w, h), ang


def poly2rbox(polys, num_cls_thata=180, radius=6.0, use_pi=False, use_gaussian=False):
    """poly (n,8) -> long-edge rbox (n,[cx cy l s θ]) [+ CSL labels] (utils/rboxs_utils.py:39-81)."""
    try:
        import cv2
        min_area_rect = cv2.minAreaRect
    except ImportError:       # (the reference imports cv2 at module import time, utils/rboxs_utils.py:6)
        min_area_rect = _min_area_rect
        global _warned_no_cv2
        if not _warned_no_cv2:
            import warnings
            warnings.warn("poly2rbox: OpenCV is not installed, the minimum-area rectangle comes from this package's own float64 "
                          "implementation (same definition; parity with cv2.minAreaRect's float32 calipers is unpinned: equal-area "
                          "ties and last-bit angles can differ)")
            _warned_no_cv2 = True
    assert polys.shape[-1] == 8
    csl_labels, rboxes = [], []
    for poly in polys:
        poly = np.float32(poly.reshape(4, 2))
        (x, y), (w, h), angle = min_area_rect(poly)
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
