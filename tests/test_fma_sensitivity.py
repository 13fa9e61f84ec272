"""CPU, build container only: how far can the ONE arithmetic difference nobody can pin move the kept set?

The reference's CUDA build is compiled by nvcc with FMA contraction on (utils/nms_rotated/src/box_iou_rotated_utils.h:262-266
comments on it); this repository and its oracle evaluate the same expressions WITHOUT contraction (-ffp-contract=off on
both sides).  SURVEY.md section 7 hard part 1(b) / VERDICT r2 item 6(a): quantify it.  oracle/_ref/libref_riou_dev_fma.so is
the reference's own header (device hull branch) compiled with -ffp-contract=fast -mfma; the greedy scan of
nms_rotated_cuda.cu:109-128 is run with it and with the uncontracted build over the bench's regimes and the thresholds
the reference's scripts use (0.1, 0.2 detect.py:217, 0.4 configs[3], 0.45 val.py --task speed), and the kept lists are
compared.  The numbers are PRINTED (pytest -s) and asserted to be small; the full N = 100k table is kept in
profiles/r3_fma_sensitivity.md (OBB_FMA_N=100000 reproduces it, ~2 minutes on 8 cores).
Not an x86 == PTX claim: gcc's contraction choices need not equal nvcc's; it measures the SIZE of the effect."""
import ctypes as C
import os

import numpy as np
import pytest

import oracle
from tests import synth

REFDIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref")
pytestmark = pytest.mark.skipif(not os.path.exists(os.path.join(REFDIR, "libref_riou_dev_fma.so")),
                                reason="oracle/_ref not built (needs /root/reference)")
f32p = np.ctypeslib.ndpointer(np.float32, flags='C')
i64p = np.ctypeslib.ndpointer(np.int64, flags='C')
THRS = (0.1, 0.2, 0.4, 0.45)
REGIMES = ("clustered_k300", "clustered_k300_18cls", "clustered_k3000", "uniform")


def _scan(tag):
    lib = C.CDLL(os.path.join(REFDIR, {"dev": "libref_riou_dev.so", "devfma": "libref_riou_dev_fma.so"}[tag]))
    fn = getattr(lib, f"ref_{tag}_nms_gt_sorted")
    fn.argtypes = [f32p, C.c_int64, C.c_float, i64p]
    fn.restype = C.c_int64
    pairs = getattr(lib, f"ref_{tag}_riou_pairs_f32")
    pairs.argtypes = [f32p, f32p, C.c_int64, f32p]

    def run(sorted_dets, thr):
        keep = np.empty(len(sorted_dets), np.int64)
        k = fn(np.ascontiguousarray(sorted_dets.reshape(-1)), len(sorted_dets), thr, keep)
        return keep[:k]
    return run, pairs


def test_uncontracted_reference_scan_equals_the_oracle(oracle_lib):
    """Pins the scan door itself: reference header, no contraction, circle cull == the oracle's kept list."""
    run, _ = _scan("dev")
    d, s = synth.s_clustered(20000, 300, 3)
    s = synth.tie_free(s)
    order = oracle.order_desc(s.numpy())
    got = order[run(d.numpy()[order], 0.4)]
    assert np.array_equal(got, oracle.nms_rotated(d.numpy(), s.numpy(), 0.4, threads=8))


def test_kept_set_flips_under_fma_contraction_are_rare(oracle_lib, capsys):
    n = int(os.environ.get("OBB_FMA_N", "20000"))
    plain, pairs_plain = _scan("dev")
    fma, pairs_fma = _scan("devfma")
    rows = []
    for regime in REGIMES:
        d, s = synth.regime_100k(regime, n)
        order = oracle.order_desc(s.numpy())
        sd = np.ascontiguousarray(d.numpy()[order])
        for thr in THRS:
            a, b = plain(sd, thr), fma(sd, thr)
            diff = len(np.setxor1d(a, b))
            rows.append((regime, thr, len(a), len(b), diff))
            assert diff <= max(4, len(a) // 500), (regime, thr, len(a), len(b), diff)      # <= 0.2 % of the kept set
    # the IoU values themselves: bits that change, largest difference (the 1e-5 scalar tolerance of the north star)
    a5, _ = synth.s_uniform(200000, 7, extent=100.0)
    b5, _ = synth.s_uniform(200000, 8, extent=100.0)
    x, y = np.ascontiguousarray(a5.numpy().reshape(-1)), np.ascontiguousarray(b5.numpy().reshape(-1))
    o0, o1 = np.empty(200000, np.float32), np.empty(200000, np.float32)
    pairs_plain(x, y, 200000, o0)
    pairs_fma(x, y, 200000, o1)
    changed = int((o0.view(np.uint32) != o1.view(np.uint32)).sum())
    maxdiff = float(np.abs(o0.astype(np.float64) - o1).max())
    assert maxdiff < 1e-5
    with capsys.disabled():
        print(f"\nFMA sensitivity of the kept set (reference header, device hull branch, gcc -ffp-contract=fast -mfma vs off), N = {n}")
        print("| regime | iou_thres | kept (no FMA) | kept (FMA) | indices that differ |")
        print("|---|---|---|---|---|")
        for r in rows:
            print(f"| {r[0]} | {r[1]} | {r[2]} | {r[3]} | {r[4]} |")
        print(f"IoU values of 200k random overlapping pairs: {changed} of 200000 change bits, max |diff| = {maxdiff:.3e}")
