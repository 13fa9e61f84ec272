"""GPU parity: DOTA Task-1 evaluation (yolov5_obb_amd/DOTA_devkit/dota_evaluation_task1.py over obb_eval_best_gt_f64)
against rec / prec / AP frozen from the reference's own voc_eval, and the per-detection (ovmax, jmax) against the oracle.
Everything is double arithmetic in a fixed order: exact equality."""
import numpy as np
import pytest

from oracle import pyref
from tests.golden.gen_golden import EVAL_CASES, EVAL_CLASSES, eval_inputs, eval_write
from tests.test_oracle_golden import G

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", list(EVAL_CASES))
def test_voc_eval_equals_the_reference_devkit(dev, oracle_lib, tmp_path, name):
    from yolov5_obb_amd.DOTA_devkit import dota_evaluation_task1 as EV
    gt, det = eval_inputs(*EVAL_CASES[name])
    detpath, annopath, imagesetfile = eval_write(str(tmp_path / "set"), gt, det)
    aps = []
    for cls in EVAL_CLASSES:
        for m07 in (True, False):
            rec, prec, ap = EV.voc_eval(detpath, annopath, imagesetfile, cls, ovthresh=0.5, use_07_metric=m07)
            assert ap == float(G[f"eval_{name}_{cls}_ap{int(m07)}"])
        assert np.array_equal(rec, G[f"eval_{name}_{cls}_rec"]) and np.array_equal(prec, G[f"eval_{name}_{cls}_prec"])
        aps.append(float(G[f"eval_{name}_{cls}_ap1"]))
    mean_ap, classaps = EV.evaluate(detpath, annopath, imagesetfile, list(EVAL_CLASSES) + ['harbor'])   # no file: skipped
    assert mean_ap == sum(aps) / 2 and np.array_equal(classaps, 100 * np.array(aps))


def test_best_gt_vs_oracle_per_detection(dev, oracle_lib):
    """Images with 0 .. 900 ground-truth quads (several 64-lane rounds), exact duplicates (ties -> first index), degenerate
    quads (NaN -> first NaN), detections far from everything (-inf, -1)."""
    from yolov5_obb_amd.DOTA_devkit.dota_evaluation_task1 import best_gt
    rng = np.random.RandomState(3)

    def quads(n, extent):
        cx, cy = rng.rand(n) * extent, rng.rand(n) * extent
        w, h, t = rng.rand(n) * 60 + 5, rng.rand(n) * 20 + 5, (rng.rand(n) - 0.5) * np.pi
        c, s = np.cos(t), np.sin(t)
        cols = []
        for sx, sy in ((1, 1), (1, -1), (-1, -1), (-1, 1)):
            cols += [cx + sx * w / 2 * c - sy * h / 2 * s, cy + sx * w / 2 * s + sy * h / 2 * c]
        return np.stack(cols, 1)
    gts, off, dets, dimg = [], [0], [], []
    for im, ng in enumerate([0, 1, 63, 64, 65, 900, 200, 0, 130]):
        g = quads(ng, 40 * np.sqrt(ng) + 50)
        if ng > 10:
            g[5] = g[2]                                           # duplicated ground truth: the first index wins
            g[7] = np.tile(g[7, :2], 4)                           # a point
            g[9, 4:] = g[9, :4]                                   # a segment walked twice
        gts.append(g); off.append(off[-1] + ng)
        nd = 40 + ng // 4
        d = quads(nd, 40 * np.sqrt(ng) + 50)
        if ng > 10:
            src = rng.randint(0, ng, nd // 2)
            d[:nd // 2] = g[src] + rng.randn(nd // 2, 8) * 1.5
            d[0] = g[2]; d[1] = g[7]; d[2] = g[9]; d[3] = g[9] + 0.25
            d[4] = d[4] + 1e6
        dets.append(d); dimg += [im] * nd
    gts, dets, dimg = np.concatenate(gts), np.concatenate(dets), np.array(dimg, dtype=np.int32)
    perm = rng.permutation(len(dets))
    dets, dimg = dets[perm], dimg[perm]
    ov, jm = best_gt(dets, dimg, gts, np.array(off))
    n_nan = n_none = 0
    for d in range(len(dets)):
        ro, rj = pyref.task1_best_gt(dets[d], gts[off[dimg[d]]:off[dimg[d] + 1]])
        if rj is None:
            assert ov[d] == -np.inf and jm[d] == -1; n_none += 1
        elif np.isnan(ro):
            assert np.isnan(ov[d]) and jm[d] == rj; n_nan += 1
        else:
            assert ov[d] == ro and jm[d] == rj, (d, ov[d], ro, jm[d], rj)
    assert n_none > 20 and len(dets) - n_none - n_nan > 200
