"""Development check of the phase-kernel NMS path (csrc/nms_mk.h) against the persistent-kernel path (csrc/nms_core.h, pinned to
the oracle by tests/test_nms_gpu.py): same kept list on every case, and the time of both.  OBB_NMS_MK is read per call.

    python tools/mk_check.py [quick]
"""
import math
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import synth  # noqa: E402
from yolov5_obb_amd import nms_rotated_ext  # noqa: E402


def run(d, s, thr, mk):
    os.environ["OBB_NMS_MK"] = str(int(mk))          # 0: persistent kernel, 1: phase kernels, 2: the library's own choice (feedback)
    return nms_rotated_ext.nms_rotated(d, s, thr)


def timed(d, s, thr, mk, reps=12):
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(d, s, thr, mk)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    return float(np.median(ts)), float(np.min(ts))


def cases(quick):
    g = torch.Generator().manual_seed(1234)
    for name in ("clustered_k300", "clustered_k300_18cls", "clustered_k3000", "uniform", "uniform_18cls", "clustered_k300_raw"):
        d, s = synth.regime_100k(name)
        yield name, d, s, 0.4
    for n in (16384, 20000, 30000, 65536, 70001):
        d, s = synth.s_uniform(n, 3)
        yield f"uniform_{n}", d, synth.tie_free(s), 0.45
        d, s = synth.s_clustered(n, 40, 5)
        yield f"clustered40_{n}", d, synth.tie_free(s), 0.3
    if quick:
        return
    n = 40000
    # identical boxes, two values of the score
    d = torch.tensor([[100.0, 100.0, 30.0, 10.0, 0.3]]).repeat(n, 1)
    yield "identical", d, synth.tie_free(torch.rand(n, generator=g)), 0.5
    # thin / tiny / huge / non-finite boxes mixed into a uniform set: brute entries and brute queries
    d, s = synth.s_uniform(n, 7)
    d[::97, 2] = 1e-4
    d[5::101, 3] = 3e-3
    d[11::503, 2:4] = 900.0
    d[13::1009, 0] = float("nan")
    d[17::1013, 1] = float("inf")
    d[19::1019, 2] = float("inf")
    d[23::1021, 4] = float("nan")
    yield "uniform_degenerate", d, synth.tie_free(s), 0.4
    # coordinates around 1e6 (coarse fp32 grid), small boxes
    d, s = synth.s_clustered(n, 500, 9)
    d[:, :2] += 1.0e6
    yield "far_origin", d, synth.tie_free(s), 0.4
    # unit-square coordinates
    d, s = synth.s_uniform(n, 11)
    d[:, :4] /= 1024.0
    yield "unit_square", d, synth.tie_free(s), 0.4
    # sizes over 2000:1
    d, s = synth.s_uniform(n, 13)
    d[:, 2:4] = torch.exp(torch.rand(n, 2, generator=g) * math.log(2000.0)) * 0.5
    yield "sizes_2000_to_1", d, synth.tie_free(s), 0.4
    # thresholds at the ends
    d, s = synth.s_clustered(n, 100, 15)
    yield "thr_0", d, synth.tie_free(s), 0.0
    yield "thr_1", d, synth.tie_free(s), 1.0
    yield "thr_0.95", d, synth.tie_free(s), 0.95
    # ties in the scores (fp16-rounded)
    d, s = synth.s_clustered(n, 200, 17)
    yield "score_ties", d, s.half().float(), 0.4
    # everything dead after the first chunk / nothing suppressed at all
    d, s = synth.s_uniform(n, 19, extent=100000.0)
    yield "sparse_nothing_suppressed", d, synth.tie_free(s), 0.4
    d, s = synth.s_clustered(n, 1, 21)
    yield "one_cluster", d, synth.tie_free(s), 0.1


def main():
    quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
    dev = torch.device("cuda:0")
    bad = 0
    for name, d, s, thr in cases(quick):
        dd, ss = d.to(dev), s.to(dev)
        ref = run(dd, ss, thr, 0).cpu().numpy()
        ok = True
        for rep in range(3):
            got = run(dd, ss, thr, 1).cpu().numpy()
            if not np.array_equal(got, ref):
                ok = False
                common = int((got[: min(len(got), len(ref))] == ref[: min(len(got), len(ref))]).sum())
                first = int(np.argmax(got[: min(len(got), len(ref))] != ref[: min(len(got), len(ref))])) if common < min(len(got), len(ref)) else -1
                print(f"  MISMATCH {name} rep {rep}: kept {len(got)} vs {len(ref)}, first difference at {first}", flush=True)
        bad += 0 if ok else 1
        t_old = timed(dd, ss, thr, 0)
        t_new = timed(dd, ss, thr, 1)
        for _ in range(3):
            got = run(dd, ss, thr, 2).cpu().numpy()
            ok = ok and np.array_equal(got, ref)
        t_auto = timed(dd, ss, thr, 2)
        print(f"{'ok ' if ok else 'BAD'} {name:28s} n {len(d):6d} kept {len(ref):6d}  persist {t_old[0]:7.3f} ms   phase kernels {t_new[0]:7.3f} ms   auto {t_auto[0]:7.3f} ms",
              flush=True)
    print("mismatching cases:", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
