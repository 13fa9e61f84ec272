"""CPU: properties of ``poly2rbox`` (utils/rboxs_utils.py:39-81 of the reference).  The function rests on cv2.minAreaRect; OpenCV
is not in this image, so parity with it cannot be pinned (bench.py: parity_unpinned).  What CAN be checked is what the function
promises whatever minAreaRect implementation it is handed: for a rectangle it returns that rectangle in the long-edge
convention -- l >= s, theta in [-pi/2, pi/2) (or the angle in [0, 180)) -- independent of the order in which the four corners
arrive, and its CSL row peaks on the angle's bin.  With OpenCV installed the same assertions run through cv2.minAreaRect and
the native rectangle is compared with it directly."""
import numpy as np
import pytest

from yolov5_obb_amd.utils import rboxs_utils as R


def _rects(n, seed):
    rng = np.random.RandomState(seed)
    l = 8 + rng.rand(n) * 400
    s = l * (0.03 + 0.9 * rng.rand(n))
    theta = (rng.rand(n) - 0.5) * R.pi * 0.9998                    # open interval: the ends of [-pi/2, pi/2) are one rectangle
    return np.stack([rng.rand(n) * 2048, rng.rand(n) * 2048, l, s, theta], 1)


def _ang_diff(a, b, period):
    d = np.abs(a - b) % period
    return np.minimum(d, period - d)


@pytest.mark.parametrize("order", ["as_is", "rolled", "reversed", "shuffled"])
def test_rectangles_come_back_in_long_edge_form(order):
    rb = _rects(3000, 1)
    poly = R.rbox2poly(rb).reshape(-1, 4, 2)
    rng = np.random.RandomState(2)
    if order == "rolled":
        poly = np.stack([np.roll(p, rng.randint(4), 0) for p in poly])
    elif order == "reversed":
        poly = poly[:, ::-1]
    elif order == "shuffled":                                    # not a ring any more: the hull is what counts
        poly = np.stack([p[rng.permutation(4)] for p in poly])
    out = R.poly2rbox(poly.reshape(-1, 8), use_pi=True)
    assert out.shape == (3000, 5)
    assert (out[:, 2] >= out[:, 3]).all()                         # long edge first
    assert (out[:, 4] >= -R.pi / 2 - 1e-9).all() and (out[:, 4] < R.pi / 2).all()
    tol = 2e-4 * (1 + np.abs(rb[:, :4]).max())                    # the corners were rounded to float32
    assert np.abs(out[:, :4] - rb[:, :4]).max() <= tol
    # the angle of a nearly square box is ill conditioned: a corner moved by eps turns the box by ~eps / (l - s)
    ang_tol = 1e-5 + 4e-4 / np.maximum(rb[:, 2] - rb[:, 3], 1e-3)
    assert (_ang_diff(out[:, 4], rb[:, 4], R.pi) <= ang_tol).all()


def test_angle_form_and_csl_row():
    rb = _rects(500, 3)
    poly = R.rbox2poly(rb)
    out_pi = R.poly2rbox(poly, use_pi=True)
    out_deg, csl = R.poly2rbox(poly, use_pi=False, use_gaussian=True, num_cls_thata=180, radius=6.0)
    assert (out_deg[:, 4] >= 0).all() and (out_deg[:, 4] < 180).all()
    assert np.allclose(out_deg[:, 4], out_pi[:, 4] * 180 / R.pi + 90, atol=1e-9)
    assert csl.shape == (500, 180)
    # gaussian_label_cpu puts the peak on bin ceil(angle) mod 180 (utils/rboxs_utils.py:24-26: index = int(90 - angle))
    peak = csl.argmax(1)
    want = (90 - (90 - out_deg[:, 4]).astype(int)) % 180
    assert (peak == want).all()
    assert np.allclose(csl.max(1), 1.0)


def test_degenerate_inputs_do_not_crash():
    pts = np.array([[5, 5, 5, 5, 5, 5, 5, 5],                       # one point
                    [0, 0, 10, 0, 20, 0, 30, 0],                    # collinear
                    [0, 0, 10, 0, 10, 10, 0, 10]], dtype=np.float64)  # axis-aligned square
    out = R.poly2rbox(pts, use_pi=True)
    assert np.isfinite(out).all()
    assert np.allclose(out[0, :4], [5, 5, 0, 0])
    assert np.allclose(out[1, :4], [15, 0, 30, 0]) and abs(_ang_diff(out[1, 4], 0.0, R.pi)) < 1e-9
    assert np.allclose(out[2, :4], [5, 5, 10, 10])


def test_native_rectangle_against_opencv_when_present():
    cv2 = pytest.importorskip("cv2")
    rng = np.random.RandomState(4)
    for _ in range(2000):
        quad = np.float32(rng.rand(4, 2) * 500)
        (cx, cy), (w, h), a = cv2.minAreaRect(quad)
        (nx, ny), (nw, nh), na = R._min_area_rect(quad)
        assert abs(w * h - nw * nh) <= 1e-3 * max(1.0, w * h)   # the same minimum area (float vs double implementation)
        if abs(w - h) > 1e-2 * max(w, h) and w * h > 1.0:
            assert abs(cx - nx) < 0.05 and abs(cy - ny) < 0.05


def test_documented_opencv_convention_on_exact_shapes():
    """cv2.minAreaRect since OpenCV 4.5.1 reports the angle in (0, 90] and the extent ALONG that direction as the width: an upright
    w x h rectangle comes back as ((cx, cy), (h, w), 90), a square standing on a corner as (.., (a, a), 45), a rectangle turned by
    t in (0, 90) degrees as (.., (w, h), t).  The native rectangle follows that convention on exact shapes (all four hull edges give
    the same rectangle: no tie to break), and the reference's long-edge normalisation behind it (utils/rboxs_utils.py:61-70) lands
    on theta = -pi/2 (angle label 0) for the upright square -- the value the CSL training labels are built from."""
    (c, (w, h), a) = R._min_area_rect(np.float32([[0, 0], [10, 0], [10, 4], [0, 4]]))
    assert np.allclose(c, (5, 2)) and np.allclose((w, h), (4, 10)) and a == 90.0
    (c, (w, h), a) = R._min_area_rect(np.float32([[0, 5], [5, 0], [10, 5], [5, 10]]))
    assert np.allclose(c, (5, 5)) and np.allclose((w, h), (50 ** 0.5, 50 ** 0.5)) and abs(a - 45.0) < 1e-9
    for t in (10.0, 30.0, 60.0, 89.0):
        r = np.deg2rad(t)
        u, v = np.array([np.cos(r), np.sin(r)]), np.array([-np.sin(r), np.cos(r)])
        quad = np.array([100 + 15 * u + 4 * v, 100 + 15 * u - 4 * v, 100 - 15 * u - 4 * v, 100 - 15 * u + 4 * v])
        (c, (w, h), a) = R._min_area_rect(quad)
        assert np.allclose(c, (100, 100), atol=1e-9) and abs(w - 30) < 1e-9 and abs(h - 8) < 1e-9 and abs(a - t) < 1e-9
    # through poly2rbox: the upright square and the upright rectangles in both orientations
    out = R.poly2rbox(np.float64([[0, 0, 10, 0, 10, 10, 0, 10], [0, 0, 10, 0, 10, 4, 0, 4], [0, 0, 4, 0, 4, 10, 0, 10]]), use_pi=True)
    assert np.allclose(out[0], [5, 5, 10, 10, -R.pi / 2])
    assert np.allclose(out[1, :4], [5, 2, 10, 4]) and abs(_ang_diff(out[1, 4], 0.0, R.pi)) < 1e-12
    assert np.allclose(out[2, :4], [2, 5, 10, 4]) and abs(_ang_diff(out[2, 4], R.pi / 2, R.pi)) < 1e-12 and out[2, 4] < 0
