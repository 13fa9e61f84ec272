"""Freezes outputs of the reference's OWN py_cpu_nms_poly and py_cpu_nms (DOTA_devkit/ResultMerge_multi_process.py:24-60,
:125-157) on seeded inputs -> tests/golden/merge_variants.npz.  Run in the build container (needs /root/reference and
oracle/_ref/libref_polyiou.so = the reference's polyiou.cpp compiled in place, standing in for its SWIG module):
    python tests/golden/gen_merge_variants.py
The inputs are regenerated from the seeds by merge_variant_inputs() in the tests; only the kept lists are stored."""
import ctypes as C
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CASES = {"a": (60, 100.0, 1), "b": (150, 200.0, 2), "c": (400, 250.0, 3), "one": (1, 10.0, 4), "two": (2, 5.0, 5)}
THRESHOLDS = (0.0, 0.2, 0.5)


def merge_variant_inputs(n, extent, seed):
    """(quads (n, 9) with two-decimal scores, horizontal boxes (n, 5) incl. zero / negative areas)."""
    rng = np.random.RandomState(seed)
    c = rng.rand(n, 2) * extent
    w, h, t = 5 + rng.rand(n) * 40, 3 + rng.rand(n) * 16, (rng.rand(n) - 0.5) * 3.1
    q = np.zeros((n, 9))
    for k, (sx, sy) in enumerate(((1, 1), (1, -1), (-1, -1), (-1, 1))):
        dx, dy = sx * w / 2, sy * h / 2
        q[:, 2 * k] = c[:, 0] + dx * np.cos(t) - dy * np.sin(t)
        q[:, 2 * k + 1] = c[:, 1] + dx * np.sin(t) + dy * np.cos(t)
    q[:, 8] = np.round(rng.rand(n), 2)
    hb = np.stack([q[:, 0], q[:, 1], q[:, 0] + rng.rand(n) * 30, q[:, 1] + rng.rand(n) * 30, q[:, 8]], 1)
    hb[::7, 2] = hb[::7, 0] - 1
    hb[3::11, 3] = hb[3::11, 1] - 5
    return q, hb


def main():
    sys.path.insert(0, ROOT)
    sys.path.insert(0, '/root/reference')
    f64p = np.ctypeslib.ndpointer(np.float64, flags='C_CONTIGUOUS')
    pol = C.CDLL(os.path.join(ROOT, 'oracle/_ref/libref_polyiou.so'))
    pol.ref_iou_poly.restype = C.c_double
    pol.ref_iou_poly.argtypes = [f64p, f64p]
    sys.modules.setdefault('shapely', types.ModuleType('shapely'))
    sys.modules['shapely.geometry'] = types.ModuleType('shapely.geometry')
    stub = types.ModuleType('DOTA_devkit.polyiou')
    stub.VectorDouble = lambda v: np.ascontiguousarray(v, dtype=np.float64)
    stub.iou_poly = lambda p, q: pol.ref_iou_poly(p, q)
    sys.modules['DOTA_devkit.polyiou'] = stub
    import DOTA_devkit
    DOTA_devkit.polyiou = stub
    import DOTA_devkit.ResultMerge_multi_process as RM
    out = {}
    for name, cfg in CASES.items():
        q, hb = merge_variant_inputs(*cfg)
        for thr in THRESHOLDS:
            with np.errstate(all='ignore'):
                out[f"{name}_poly_all_{thr}"] = np.array(RM.py_cpu_nms_poly(q.copy(), thr), dtype=np.int64)
                out[f"{name}_hbb9_{thr}"] = np.array(RM.py_cpu_nms(q.copy(), thr), dtype=np.int64)
                out[f"{name}_hbb5_{thr}"] = np.array(RM.py_cpu_nms(hb.copy(), thr), dtype=np.int64)
    np.savez_compressed(os.path.join(HERE, 'merge_variants.npz'), **out)
    print(f"wrote tests/golden/merge_variants.npz: {len(out)} kept lists")


if __name__ == "__main__":
    main()
