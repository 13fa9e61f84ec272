"""TEST INFRASTRUCTURE -- CPU oracle for the oriented-box hot path.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import this package.  The product (``yolov5_obb_amd``) never
does: it fails loudly when its HIP library is missing instead of falling back.

Contents
--------
* ``liboracle.so``  (``obb_oracle.c`` + ``*_impl.inc``): plain-C restatement of
  the reference's native algorithms (rotated IoU, quad IoU, greedy NMS).
* ``pyref.py``: numpy / torch-CPU restatement of the reference's Python hot
  path (``non_max_suppression_obb``, CSL encode, ``ComputeLoss`` ...).
* ``_ref/``  (git-ignored, built by ``oracle/Makefile`` when ``/root/reference``
  exists): the reference's *own* sources compiled in place; used to pin the
  restatement and as the ``"reference"`` CPU baseline.

Parity status: pinned (see ``tests/test_oracle_vs_ref.py`` and ``tests/golden``).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

_f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
_f64p = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
_i64p = np.ctypeslib.ndpointer(dtype=np.int64, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")


def build(with_ref=True):
    """Compile liboracle.so (and _ref/ when the reference tree is present)."""
    target = "all" if with_ref else "oracle"
    subprocess.run(["make", "-s", "-C", _HERE, target], check=True)


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    path = os.path.join(_HERE, "liboracle.so")
    if not os.path.exists(path):
        build(with_ref=False)
    L = C.CDLL(path)
    L.oracle_riou_f32.restype = C.c_float
    L.oracle_riou_f32.argtypes = [_f32p, _f32p]
    L.oracle_riou_f64.restype = C.c_double
    L.oracle_riou_f64.argtypes = [_f64p, _f64p]
    L.oracle_piou_f32.restype = C.c_float
    L.oracle_piou_f32.argtypes = [_f32p, _f32p]
    L.oracle_piou_f64.restype = C.c_double
    L.oracle_piou_f64.argtypes = [_f64p, _f64p]
    L.oracle_order_desc_f32.restype = None
    L.oracle_order_desc_f32.argtypes = [_f32p, C.c_int64, _i64p]
    L.oracle_set_threads.restype = None
    L.oracle_set_threads.argtypes = [C.c_int]
    L.oracle_nms_rotated_f32.restype = C.c_int64
    L.oracle_nms_rotated_f32.argtypes = [_f32p, _f32p, C.c_int64, C.c_float, C.c_int, _i64p]
    L.oracle_nms_rotated_f64.restype = C.c_int64
    L.oracle_nms_rotated_f64.argtypes = [_f64p, _f64p, C.c_int64, C.c_double, C.c_int, _i64p]
    L.oracle_nms_poly_f32.restype = C.c_int64
    L.oracle_nms_poly_f32.argtypes = [_f32p, C.c_int64, C.c_float, _i64p]
    L.oracle_devkit_poly_nms.restype = C.c_int
    L.oracle_devkit_poly_nms.argtypes = [_i32p, _f32p, C.c_int, C.c_int, C.c_float]
    L.oracle_riou_matrix_f32.restype = None
    L.oracle_riou_matrix_f32.argtypes = [_f32p, C.c_int64, _f32p, C.c_int64, _f32p]
    L.oracle_piou_matrix_f32.restype = None
    L.oracle_piou_matrix_f32.argtypes = [_f32p, C.c_int64, C.c_int64, _f32p, C.c_int64, C.c_int64, _f32p]
    L.oracle_piou_matrix_f64.restype = None
    L.oracle_piou_matrix_f64.argtypes = [_f64p, C.c_int64, _f64p, C.c_int64, _f64p]
    L.oracle_devkit_overlaps.restype = None
    L.oracle_devkit_overlaps.argtypes = [_f32p, _f32p, _f32p, C.c_int, C.c_int]
    L.oracle_rbox2quad_devkit.restype = None
    L.oracle_rbox2quad_devkit.argtypes = [_f32p, _f32p]
    _LIB = L
    return L


def _c(a, dt):
    return np.ascontiguousarray(np.asarray(a, dtype=dt))


# ---- convenience wrappers (numpy in, numpy out) ---------------------------------
def riou_pairs(a5, b5):
    """Element-wise rotated IoU of two (n,5) arrays (dtype float32 or float64)."""
    a5 = np.asarray(a5)
    dt = np.float64 if a5.dtype == np.float64 else np.float32
    a5, b5 = _c(a5, dt), _c(b5, dt)
    f = lib().oracle_riou_f64 if dt == np.float64 else lib().oracle_riou_f32
    return np.array([f(a5[i], b5[i]) for i in range(len(a5))], dtype=dt)


def riou_matrix(a5, b5):
    a5, b5 = _c(a5, np.float32), _c(b5, np.float32)
    out = np.empty((len(a5), len(b5)), np.float32)
    lib().oracle_riou_matrix_f32(a5.reshape(-1), len(a5), b5.reshape(-1), len(b5), out.reshape(-1))
    return out


def piou_matrix(a, b):
    """Quad IoU matrix; rows may be wider than 8 (only the first 8 columns are read)."""
    a, b = np.asarray(a), np.asarray(b)
    if a.dtype == np.float64:
        a, b = _c(a[:, :8], np.float64), _c(b[:, :8], np.float64)
        out = np.empty((len(a), len(b)), np.float64)
        lib().oracle_piou_matrix_f64(a.reshape(-1), len(a), b.reshape(-1), len(b), out.reshape(-1))
        return out
    a, b = _c(a, np.float32), _c(b, np.float32)
    out = np.empty((len(a), len(b)), np.float32)
    lib().oracle_piou_matrix_f32(a.reshape(-1), len(a), a.shape[1], b.reshape(-1), len(b), b.shape[1], out.reshape(-1))
    return out


def order_desc(scores):
    s = _c(scores, np.float32)
    out = np.empty(len(s), np.int64)
    lib().oracle_order_desc_f32(s, len(s), out)
    return out


def nms_rotated(dets, scores, thr, ge=False, threads=1):
    """Greedy rotated NMS.  ge=False: CUDA semantics (IoU > thr); ge=True: CPU semantics (>=).
    threads > 1: the inner loop (one kept box against the later ones) runs on that many OpenMP threads -- same result
    for any thread count (obb_oracle.c: oracle_set_threads)."""
    dets = np.asarray(dets)
    lib().oracle_set_threads(int(threads))
    if dets.dtype == np.float64:
        d, s = _c(dets, np.float64), _c(scores, np.float64)
        keep = np.empty(len(d), np.int64)
        # the reference's kernels take `const float iou_threshold` (nms_rotated_cuda.cu:15, nms_rotated_cpu.cpp:12): the
        # double IoU is compared with the FLOAT value of the threshold
        k = lib().oracle_nms_rotated_f64(d.reshape(-1), s, len(d), float(np.float32(thr)), int(ge), keep)
        return keep[:k].copy()
    d, s = _c(dets, np.float32), _c(scores, np.float32)
    keep = np.empty(len(d), np.int64)
    k = lib().oracle_nms_rotated_f32(d.reshape(-1), s, len(d), float(thr), int(ge), keep)
    return keep[:k].copy()


def nms_poly(polys9, thr):
    p = _c(polys9, np.float32)
    keep = np.empty(len(p), np.int64)
    k = lib().oracle_nms_poly_f32(p.reshape(-1), len(p), float(thr), keep)
    return keep[:k].copy()


def devkit_poly_nms(sorted_polys, thr):
    p = _c(sorted_polys, np.float32)
    keep = np.empty(len(p), np.int32)
    k = lib().oracle_devkit_poly_nms(keep, p.reshape(-1), len(p), p.shape[1] if p.ndim == 2 else 9, float(thr))
    return keep[:k].copy()


def devkit_overlaps(boxes, query):
    b, q = _c(boxes, np.float32), _c(query, np.float32)
    out = np.zeros((len(b), len(q)), np.float32)
    lib().oracle_devkit_overlaps(out.reshape(-1), b.reshape(-1), q.reshape(-1), len(b), len(q))
    return out
