"""GPU, the whole offline OBB-mAP pipeline in ONE test (the stand-in for north_star's "OBB mAP within 0.1" clause; weights and
DOTA are not available offline): Detect decode -> non_max_suppression_obb -> val.py's polygons in native tile coordinates
(val.py:226-236) -> the JSON record's roundings (val.py:61-66) -> Task1_<class>.txt lines (tools/TestJson2VocClassTxt.py:39-47)
-> tile -> full-image merge with polygon NMS (DOTA_devkit/ResultMerge_multi_process.py:175-236) -> DOTA Task-1 voc_eval
(DOTA_devkit/dota_evaluation_task1.py:88-249), every stage from this package on the GPU, against the same chain built from
the oracle on the CPU.  Both chains start from the same raw head outputs and are scored against the same ground-truth files.

What may differ: the decoded boxes differ by a few fp32 ulps between the device's and the host's libm (tests/test_head_gpu.py),
so a coordinate that sits within ~1e-4 px of a rounding boundary of the 0.1 px text format can land on the other side, and a
confidence within an ulp of a 1e-5 boundary likewise.  The text files are therefore compared token by token (names exact,
numbers within one step of their format); rec / prec / AP are compared exactly -- they only move if such a flip crosses the
0.5 IoU or the 0.2 merge threshold or swaps two scores, which the seeded inputs do not do."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

from oracle import pyref
from tests import synth
from tests.test_e2e_gpu import _planted_head

pytestmark = pytest.mark.gpu

CLASSES = ['plane', 'baseball-diamond', 'bridge', 'ground-track-field', 'small-vehicle', 'large-vehicle', 'ship', 'tennis-court',
           'basketball-court', 'storage-tank', 'soccer-ball-field', 'roundabout', 'harbor', 'swimming-pool', 'helicopter',
           'container-crane']                                  # tools/TestJson2VocClassTxt.py:24-25 (DOTA-v1.5)
TILES = [(0, 0), (824, 0), (0, 824)]                           # 1024 px tiles, 200 px overlap
GAIN, PAD = 0.25, (0.0, 0.0)                                   # a 256 px model input per 1024 px tile


def _task1_lines(tile_names, polyn_per_tile):
    """val.py:61-66 (score rounded to 5, polygon to 1 decimal) + TestJson2VocClassTxt.py:39-47 ("%s" of the rounded floats)."""
    out = {}
    for name, polyn in zip(tile_names, polyn_per_tile):
        for p in polyn.tolist():
            score = round(p[-2], 5)
            poly = [round(x, 1) for x in p[:8]]
            line = "%s %s %s %s %s %s %s %s %s %s" % (name, score, poly[0], poly[1], poly[2], poly[3], poly[4], poly[5], poly[6], poly[7])
            out.setdefault(CLASSES[int(p[-1])], []).append(line)
    return out


def _same_up_to_one_rounding_step(a_lines, b_lines):
    assert len(a_lines) == len(b_lines)
    flips = total = 0
    for la, lb in zip(a_lines, b_lines):
        ta, tb = la.split(' '), lb.split(' ')
        assert ta[0] == tb[0] and len(ta) == len(tb) == 10, (la, lb)
        assert abs(float(ta[1]) - float(tb[1])) <= 1.001e-2, (la, lb)          # merged files carry 2 decimals, tile files 5
        for x, y in zip(ta[2:], tb[2:]):
            total += 1
            if x != y:
                flips += 1
                assert abs(float(x) - float(y)) <= 0.1001, (la, lb)
    assert flips <= max(2, total // 50), (flips, total)
    return flips


def test_detect_to_task1_ap_chain(dev, oracle_lib, tmp_path):
    from yolov5_obb_amd import _lib
    from yolov5_obb_amd.DOTA_devkit import ResultMerge_multi_process as RM
    from yolov5_obb_amd.DOTA_devkit import dota_evaluation_task1 as EV
    from yolov5_obb_amd.utils.general import non_max_suppression_obb
    from yolov5_obb_amd.val import val_postprocess
    nc, sizes, na = 16, (32, 16, 8), 3
    no = 5 + nc + 180
    origs = ["P0001", "P0002", "P0007"]
    tile_names = [f"{o}__1__{x}___{y}" for o in origs for (x, y) in TILES]
    bs = len(tile_names)
    raw = _planted_head(bs, na, no, sizes, nc, seed=41)
    anchors = synth.grid_anchors()
    kw = dict(conf_thres=0.6, iou_thres=0.4, multi_label=True, max_det=1000)

    # ---- the oracle chain (CPU)
    z_ref = pyref.detect_decode(raw, anchors, synth.DEFAULT_STRIDES)
    det_ref = pyref.non_max_suppression_obb(z_ref.clone(), **kw)
    polyn_ref = [pyref.val_postprocess(d.clone(), GAIN, PAD)[2] for d in det_ref]
    lines_ref = _task1_lines(tile_names, polyn_ref)
    merged_ref = {c: pyref.merge_result_lines(l, 0.2) for c, l in lines_ref.items()}

    # ---- ground truth, from the oracle's merged detections: two of three are objects (shifted by 1.5 px, some difficult),
    # every third one is a false positive; plus objects nobody detected
    rng = np.random.RandomState(5)
    gt = {o: [] for o in origs}
    k = 0
    for c in sorted(merged_ref):
        for line in merged_ref[c]:
            tok = line.split(' ')
            k += 1
            if k % 3 == 0:
                continue
            q = [float(v) + 1.5 for v in tok[2:]]
            gt[tok[0]].append(' '.join(f"{v:.1f}" for v in q) + f" {c} {int(k % 7 == 0)}")
    for o in origs:
        for _ in range(3):
            cx, cy = rng.rand(2) * 1500 + 3000                       # far from every detection
            gt[o].append(' '.join(f"{v:.1f}" for v in (cx, cy, cx + 40, cy, cx + 40, cy + 20, cx, cy + 20)) + f" {CLASSES[int(rng.randint(0, nc))]} 0")
    anno = tmp_path / "labelTxt"; anno.mkdir()
    for o in origs:
        (anno / f"{o}.txt").write_text('\n'.join(gt[o]) + '\n')
    (tmp_path / "imgnamefile.txt").write_text('\n'.join(origs) + '\n')
    gt_by_image = {o: EV.parse_gt(str(anno / f"{o}.txt")) for o in origs}

    # ---- this package (GPU): decode kernel, fused NMS, val tail, merge, evaluation
    a_total = sum(na * n * n for n in sizes)
    z = torch.empty((bs, a_total, no), device=dev)
    off = 0
    for i, r in enumerate(raw):
        conv = r.permute(0, 1, 4, 2, 3).contiguous().view(bs, na * no, r.shape[2], r.shape[3]).to(dev)
        px = (anchors[i] * synth.DEFAULT_STRIDES[i]).reshape(-1).tolist()
        arr = (C.c_float * len(px))(*px)
        rc = _lib.lib().obb_detect_decode(_lib.ptr(conv), 0, bs, na, no, r.shape[2], r.shape[3], C.cast(arr, C.c_void_p),
                                          float(synth.DEFAULT_STRIDES[i]), None, _lib.ptr(z), a_total, off, _lib.stream_ptr(dev))
        assert rc == 0
        off += na * r.shape[2] * r.shape[3]
    det = non_max_suppression_obb(z, **kw)
    assert [int(d.shape[0]) for d in det] == [int(d.shape[0]) for d in det_ref] and sum(int(d.shape[0]) for d in det) >= 60
    polyn = [val_postprocess(d, ratio_pad=((GAIN, GAIN), PAD))[2].cpu() for d in det]
    lines = _task1_lines(tile_names, polyn)
    assert sorted(lines) == sorted(lines_ref) and len(lines) >= 6
    src, dst = tmp_path / "src", tmp_path / "dst"
    src.mkdir()
    flips = 0
    for c in lines:
        flips += _same_up_to_one_rounding_step(lines[c], lines_ref[c])
        (src / f"Task1_{c}.txt").write_text('\n'.join(lines[c]) + '\n')
    RM.mergebypoly(str(src), str(dst))

    aps, aps_ref, n_merged = [], [], 0
    for c in sorted(lines):
        got_lines = (dst / f"Task1_{c}.txt").read_text().splitlines()
        flips += _same_up_to_one_rounding_step(got_lines, merged_ref[c])
        n_merged += len(got_lines)
        for m07 in (True, False):
            rec, prec, ap = EV.voc_eval(str(dst / "Task1_{:s}.txt"), str(anno / "{:s}.txt"), str(tmp_path / "imgnamefile.txt"), c,
                                        ovthresh=0.5, use_07_metric=m07)
            rrec, rprec, rap = pyref.task1_voc_eval(gt_by_image, origs, merged_ref[c], c, ovthresh=0.5, use_07_metric=m07)
            assert abs(ap - rap) <= 1e-12, (c, m07, ap, rap)
            assert np.array_equal(rec, rrec) and np.array_equal(prec, rprec), c
        aps.append(ap); aps_ref.append(rap)
    # the merge removed the duplicates of the tile overlaps, the evaluation is not trivial (neither 0 nor 1 everywhere)
    assert n_merged < sum(len(v) for v in lines.values())
    m_ap, m_ref = float(np.mean(aps)), float(np.mean(aps_ref))
    assert abs(m_ap - m_ref) <= 1e-12 and 0.05 < m_ref < 0.999, (m_ap, m_ref)
    print(f"chain: {sum(len(v) for v in lines.values())} tile detections -> {n_merged} merged, mAP {m_ap:.6f} (oracle {m_ref:.6f}), "
          f"{flips} text tokens one rounding step apart")
