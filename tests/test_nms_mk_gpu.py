"""GPU parity of the phase-kernel path of the single-list rotated NMS (yolov5_obb_amd/csrc/nms_mk.h: one kernel per phase, query-
centric probes of a per-step spatial hash) -- forced on with OBB_NMS_MK=1 -- against the CPU oracle and against the persistent
kernel of csrc/nms_core.h (OBB_NMS_MK=0; itself pinned to the oracle at N = 100,000 by tests/test_nms_gpu.py).

What the path must get right beyond the plain case: brute entries and brute queries (boxes that are not finite, ill conditioned
against the extent of the data, or far larger than the rest), score ties, thresholds at the ends, and every way it hands a call
over to the persistent kernel behind it -- too few enqueued steps (OBB_NMS_MK_STEPS), a chunk full of brute boxes, a pending list
that cannot hold the undecided pairs.  Same kept indices, same order: nms_rotated_cuda.cu:60 (strict >), :109-128 (the scan)."""
import math
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from tests import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(d, s, thr, mode, xlds=None):
    """mode: OBB_NMS_MK, read per call by the library: 0 persistent kernel, 1 phase kernels, 2 its own choice.
    xlds: OBB_NMS_MK_XLDS, also read per call: 1 = the cross probe on a table of the kept rows in LDS (k_mk_cross_lds), 0 = on the
    chunk's table in global memory (k_mk_probe<true>), None = the library's choice (by what the previous call kept)."""
    from yolov5_obb_amd import nms_rotated_ext
    old = {k: os.environ.get(k) for k in ("OBB_NMS_MK", "OBB_NMS_MK_XLDS")}
    os.environ["OBB_NMS_MK"] = str(mode)
    if xlds is None:
        os.environ.pop("OBB_NMS_MK_XLDS", None)
    else:
        os.environ["OBB_NMS_MK_XLDS"] = str(xlds)
    try:
        return nms_rotated_ext.nms_rotated(d, s, thr).cpu().numpy()
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


@pytest.mark.parametrize("regime", ["clustered_k300", "clustered_k300_raw", "clustered_k300_18cls", "clustered_k3000", "clustered_k3000_18cls", "uniform",
                                    "uniform_18cls"])
def test_full_size_100k_phase_kernels_equal_persistent_kernel(dev, regime):
    """BASELINE configs[3] size, the regimes bench.py times: three forced runs of the phase kernels (their step estimate comes
    from the previous run: the first run ends in the persistent kernel's hands or not, depending on the regime) and three runs
    of the library's own choice, against the persistent kernel's list (test_nms_gpu.py::test_full_size_100k_exact pins that one
    to the oracle)."""
    dets, scores = synth.regime_100k(regime)
    d, s = dets.to(dev), scores.to(dev)
    ref = _run(d, s, 0.4, 0)
    for mode, xlds in ((1, 0), (1, 0), (1, 1), (1, 1), (1, None), (2, None), (2, None), (2, None)):
        got = _run(d, s, 0.4, mode, xlds)
        assert len(got) == len(ref) and np.array_equal(got, ref), (regime, mode, xlds, len(got), len(ref))


def _special_cases():
    g = torch.Generator().manual_seed(1234)
    n = 24000
    d = torch.tensor([[100.0, 100.0, 30.0, 10.0, 0.3]]).repeat(n, 1)
    yield "identical", d, synth.tie_free(torch.rand(n, generator=g)), 0.5
    d, s = synth.s_uniform(n, 7)
    d[::97, 2] = 1e-4                       # thin / tiny / huge / non-finite boxes: brute entries and brute queries
    d[5::101, 3] = 3e-3
    d[11::503, 2:4] = 900.0
    d[13::1009, 0] = float("nan")
    d[17::1013, 1] = float("inf")
    d[19::1019, 2] = float("inf")
    d[23::1021, 4] = float("nan")
    yield "degenerate_many", d, synth.tie_free(s), 0.4
    d, s = synth.s_uniform(n, 8)
    d[7, 2] = 2e-4; d[4000, 3] = 1e-3; d[9000, 2:4] = 700.0; d[15000, 0] = float("nan"); d[20000, 3] = float("inf")
    yield "degenerate_few", d, synth.tie_free(s), 0.4          # a handful of stray boxes: stays on the phase kernels
    d, s = synth.s_clustered(n, 500, 9)
    d[:, :2] += 1.0e6
    yield "far_origin", d, synth.tie_free(s), 0.4
    d, s = synth.s_uniform(n, 11)
    d[:, :4] /= 1024.0
    yield "unit_square", d, synth.tie_free(s), 0.4
    d, s = synth.s_uniform(n, 13)
    d[:, 2:4] = torch.exp(torch.rand(n, 2, generator=g) * math.log(2000.0)) * 0.5
    yield "sizes_2000_to_1", d, synth.tie_free(s), 0.4
    d, s = synth.s_clustered(n, 100, 15)
    yield "thr_0", d, synth.tie_free(s), 0.0
    yield "thr_1", d, synth.tie_free(s), 1.0
    d, s = synth.s_clustered(n, 200, 17)
    yield "score_ties", d, s.half().float(), 0.4
    d, s = synth.s_uniform(n, 19, extent=100000.0)
    yield "wide_nothing_suppressed", d, synth.tie_free(s), 0.4
    d, s = synth.s_clustered(n, 1, 21)
    yield "one_cluster", d, synth.tie_free(s), 0.1
    d, s = synth.s_uniform(16384, 23)                           # the smallest list the path takes
    yield "n_16384", d, synth.tie_free(s), 0.45
    d, s = synth.s_uniform(16385, 24)
    dd, _ = synth.with_classes(d, 5, 3)
    yield "n_16385_classes", dd, synth.tie_free(s), 0.45


_SPECIAL = {name: (d, s, thr) for name, d, s, thr in _special_cases()}


@pytest.mark.parametrize("name", list(_SPECIAL))
def test_special_inputs_against_the_oracle(dev, oracle_lib, name):
    import oracle
    d, s, thr = _SPECIAL[name]
    ref = oracle.nms_rotated(d.numpy(), s.numpy(), thr, threads=min(os.cpu_count() or 1, 32))
    dd, ss = d.to(dev), s.to(dev)
    for mode, xlds in ((1, 0), (1, 0), (1, 1), (1, 1), (0, None)):
        got = _run(dd, ss, thr, mode, xlds)
        assert len(got) == len(ref) and np.array_equal(got, ref), (name, mode, xlds, len(got), len(ref))


_HANDOVER = r"""
import os, sys
import numpy as np, torch
sys.path.insert(0, sys.argv[1])
from tests import synth
from yolov5_obb_amd import nms_rotated_ext
dev = torch.device("cuda:0")
for name in ("uniform", "clustered_k3000"):
    d, s = synth.regime_100k(name, 60000)
    d, s = d.to(dev), s.to(dev)
    os.environ["OBB_NMS_MK"] = "0"
    ref = nms_rotated_ext.nms_rotated(d, s, 0.4).cpu().numpy()
    os.environ["OBB_NMS_MK"] = "1"
    for rep in range(4):
        os.environ["OBB_NMS_MK_XLDS"] = str(rep & 1)
        got = nms_rotated_ext.nms_rotated(d, s, 0.4).cpu().numpy()
        assert np.array_equal(got, ref), (name, rep, len(got), len(ref))
print("handover ok")
"""


@pytest.mark.parametrize("pend", [300, 5000])
def test_pending_list_overflow_hands_over(dev, pend):
    """OBB_NMS_MK_PEND shrinks the list of pairs the quick tests leave undecided: it overflows in the first pair phase (300) or
    in a cross phase (5000), the phase kernels stand back (bail) and the persistent kernel redoes the stage in flight -- the
    chunk from its first member, or the cross phase of the rows that are kept already."""
    env = dict(os.environ, OBB_NMS_MK_PEND=str(pend))
    r = subprocess.run([sys.executable, "-c", _HANDOVER, ROOT], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "handover ok" in r.stdout, (r.stdout[-400:], r.stderr[-1500:])


@pytest.mark.parametrize("steps", [1, 2, 3])
def test_too_few_enqueued_steps_are_finished_by_the_persistent_kernel(dev, steps):
    """OBB_NMS_MK_STEPS (read once per process, hence the subprocess) pins the number of enqueued steps: with 1, 2 or 3 the call
    is still in flight -- a chunk selected, or its rows kept and the next chunk selected -- when the persistent kernel behind the
    steps takes over from the control block (NmsResume, csrc/nms_core.h)."""
    env = dict(os.environ, OBB_NMS_MK_STEPS=str(steps))
    r = subprocess.run([sys.executable, "-c", _HANDOVER, ROOT], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "handover ok" in r.stdout, (r.stdout[-400:], r.stderr[-1500:])
