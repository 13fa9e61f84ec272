"""Bounded runs of the random differential checks (tools/self_fuzz.py: the fused non_max_suppression_obb on random shapes, self-sorting
segments against the sort-kernel path and against the oracle restatement of utils/general.py:772-862; tools/nms_fuzz.py: the single-list
NMS entry points against the C oracle).  The long runs of round 6 (2,240 + 470 + 580 cases) found one difference in the product -- label
rows passed the class filter, fixed in csrc/nmsobb_impl.h: k_append_extra -- and one that no CPU oracle can pin (float64 IoU of exact
duplicates against thr = 1.0: the last bit of the platform's double sin / cos)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(tool, *args, env=None):
    e = dict(os.environ)
    e.update(env or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", tool), *args], capture_output=True, text=True, timeout=900, env=e, cwd=ROOT)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0, tail
    return r.stdout, tail


def test_fused_driver_on_random_shapes(dev, oracle_lib):
    out, tail = _run("self_fuzz.py", "21", "40", env={"FUZZ_ORACLE_EVERY": "2"})
    assert "self_fuzz: 40 cases, 0 mismatches" in out, tail


def test_single_list_nms_on_random_inputs(dev, oracle_lib):
    out, tail = _run("nms_fuzz.py", "21", "20", env={"FUZZ_SECONDS": "80"})
    assert ", 0 mismatches" in out and "nms_fuzz seed 21" in out, tail


def test_pairwise_iou_on_random_inputs(dev, oracle_lib):
    """tools/iou_fuzz.py: rotated and quad IoU matrices on random sizes / extents up to 60000 / shapes (identical boxes, zero areas,
    reversed rings): quads bit for bit (the proved and the searched skip rules included), rotated within the documented last-bit bar."""
    out, tail = _run("iou_fuzz.py", "21", "40")
    assert "iou_fuzz seed 21: 40 cases, 0 mismatches" in out, tail
