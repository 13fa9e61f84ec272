"""GPU parity: rotated / quad NMS and pairwise IoU through the C ABI vs the CPU oracle.

Bar: kept-index sequences bit-exact (same indices, same order); IoU values bit-exact
for the rotated/quad kernels (same IEEE fp32 operations as the oracle); the devkit overlaps
too, up to a last-bit difference of the two double libms' cos / sin (<= 1e-5 abs).
"""
import os

import numpy as np
import pytest
import torch

import oracle
from tests import synth

pytestmark = pytest.mark.gpu


def _gpu_keep(dets, scores, thr, dev):
    from yolov5_obb_amd import nms_rotated_ext
    k = nms_rotated_ext.nms_rotated(dets.to(dev), scores.to(dev), thr)
    return k.cpu().numpy()


@pytest.mark.parametrize("n,seed", [(1, 0), (2, 0), (63, 1), (64, 1), (65, 2), (1000, 0), (2048, 3), (2049, 3), (5000, 1)])
@pytest.mark.parametrize("thr", [0.1, 0.4])
def test_nms_rotated_uniform(dev, oracle_lib, n, seed, thr):
    dets, scores = synth.s_uniform(n, seed)
    scores = synth.tie_free(scores)
    ref = oracle.nms_rotated(dets.numpy(), scores.numpy(), thr)
    got = _gpu_keep(dets, scores, thr, dev)
    assert np.array_equal(ref, got)


@pytest.mark.parametrize("n,k,seed", [(1000, 30, 0), (10000, 300, 1), (30000, 300, 2)])
@pytest.mark.parametrize("thr", [0.2, 0.45])
def test_nms_rotated_clustered(dev, oracle_lib, n, k, seed, thr):
    dets, scores = synth.s_clustered(n, k, seed)
    scores = synth.tie_free(scores)
    ref = oracle.nms_rotated(dets.numpy(), scores.numpy(), thr)
    got = _gpu_keep(dets, scores, thr, dev)
    assert np.array_equal(ref, got)


def test_nms_rotated_class_offsets(dev, oracle_lib):
    dets, scores = synth.s_clustered(6000, 100, 5)
    dets, _ = synth.with_classes(dets, 16, 5)
    scores = synth.tie_free(scores)
    ref = oracle.nms_rotated(dets.numpy(), scores.numpy(), 0.4)
    got = _gpu_keep(dets, scores, 0.4, dev)
    assert np.array_equal(ref, got)


def test_nms_rotated_ties_stable(dev, oracle_lib):
    """fp16-rounded scores: many ties; documented rule = ascending original index among equals."""
    dets, scores = synth.s_clustered(4000, 60, 11)
    scores = scores.half().float()
    assert scores.unique().numel() < scores.numel()
    ref = oracle.nms_rotated(dets.numpy(), scores.numpy(), 0.3)
    got = _gpu_keep(dets, scores, 0.3, dev)
    assert np.array_equal(ref, got)


def test_nms_rotated_thresholds_edge(dev, oracle_lib):
    dets, scores = synth.s_clustered(1500, 40, 3)
    scores = synth.tie_free(scores)
    for thr in (0.0, 1.0, -0.5):
        ref = oracle.nms_rotated(dets.numpy(), scores.numpy(), thr)
        got = _gpu_keep(dets, scores, thr, dev)
        assert np.array_equal(ref, got), thr


def test_nms_rotated_degenerate_boxes(dev, oracle_lib):
    dets, scores = synth.s_uniform(500, 4)
    dets[::7, 2] = 0.0            # zero width: area < 1e-14 -> IoU 0 with everything
    dets[::11, 3] = 1e-9
    dets[5:200:13] = dets[4:199:13]   # exact duplicates
    scores = synth.tie_free(scores)
    ref = oracle.nms_rotated(dets.numpy(), scores.numpy(), 0.3)
    got = _gpu_keep(dets, scores, 0.3, dev)
    assert np.array_equal(ref, got)


@pytest.mark.parametrize("variant", ["needles", "far_outlier", "non_finite", "dropped_small"])
def test_index_path_with_degenerate_boxes(dev, oracle_lib, variant):
    """The indexed cross phase (n >= 8192 and a step that keeps >= 512 rows) with the boxes it must keep OUT of the index:
    needle boxes (ill-conditioned far pairs: brute list), a far outlier that stretches the extent until most boxes are
    brute (the build gives up after its counting pass), non-finite coordinates / sizes, oversized boxes, and boxes
    dropped by the small-box filter.  Same kept list as the oracle in every case, twice (the build order is not fixed)."""
    import os
    from yolov5_obb_amd import _lib, nms_rotated_ext
    n = 24000
    dets, scores = synth.s_uniform(n, 21, extent=2048.0)
    g = torch.Generator().manual_seed(5)
    pick = torch.randperm(n, generator=g)
    flags = 0
    if variant == "needles":
        dets[pick[:300], 2] = torch.rand(300, generator=g) * 0.05 + 0.002       # short side far below the conditioning bound
        dets[pick[300:330], 2:4] = torch.tensor([3000.0, 900.0])                # larger than the data: too large for the top level
        dets[pick[330:400]] = dets[pick[400:470]]                               # exact duplicates
    elif variant == "far_outlier":
        dets[pick[0], 0] = 3.0e6
        dets[pick[1], 1] = -2.0e6
    elif variant == "non_finite":
        dets[pick[:20], 0] = float("nan")
        dets[pick[20:40], 1] = float("inf")
        dets[pick[40:60], 2] = float("inf")
        dets[pick[60:80], 4] = float("nan")
        dets[pick[80:100], 2:4] = 0.0
    else:
        dets[pick[:2000], 3] = 0.0005                                           # obb_nms drops them (min side < 0.001)
        flags = _lib.OBB_NMS_DROP_SMALL
    scores = synth.tie_free(scores)
    d, s = dets.to(dev), scores.to(dev)
    if variant == "dropped_small":
        keep_mask = dets[:, 2:4].min(1)[0] >= 0.001
        idx = torch.nonzero(keep_mask).squeeze(1).numpy()
        ref = idx[oracle.nms_rotated(dets[keep_mask].numpy(), scores[keep_mask].numpy(), 0.4, threads=min(os.cpu_count() or 1, 32))]
    else:
        ref = oracle.nms_rotated(dets.numpy(), scores.numpy(), 0.4, threads=min(os.cpu_count() or 1, 32))
    assert len(ref) > 5000                                                      # chunks keep thousands of rows: the indexed form runs
    for _ in range(2):
        got = nms_rotated_ext._run_rotated(d, s, 0.4, flags=flags).cpu().numpy()
        assert np.array_equal(ref, got), (variant, len(ref), len(got))


def test_obb_nms_wrapper_small_box_filter(dev, oracle_lib):
    """obb_nms drops boxes with min(w,h) < 0.001 before NMS (nms_rotated_wrapper.py:32-39)."""
    from yolov5_obb_amd.utils.nms_rotated import obb_nms
    dets, scores = synth.s_clustered(3000, 50, 8)
    dets[::5, 2] = 0.0005
    scores = synth.tie_free(scores)
    valid = (dets[:, 2:4].min(1)[0] >= 0.001).numpy()
    idx = np.nonzero(valid)[0]
    ref = idx[oracle.nms_rotated(dets.numpy()[valid], scores.numpy()[valid], 0.4)]
    kept, inds = obb_nms(dets.to(dev), scores.to(dev), 0.4)
    assert inds.device.type == "cuda" and inds.dtype == torch.int64
    assert np.array_equal(ref, inds.cpu().numpy())
    assert torch.equal(kept.cpu(), dets[inds.cpu()])
    # all too small -> nothing
    d2 = dets.clone(); d2[:, 3] = 0.0
    _, inds2 = obb_nms(d2.to(dev), scores.to(dev), 0.4)
    assert inds2.numel() == 0
    # empty input
    _, inds3 = obb_nms(dets[:0].to(dev), scores[:0].to(dev), 0.4)
    assert inds3.numel() == 0 and inds3.dtype == torch.int64
    # numpy in -> numpy indices out
    k4, inds4 = obb_nms(dets.numpy(), scores.numpy(), 0.4, device_id=0)
    assert isinstance(inds4, np.ndarray) and np.array_equal(inds4, ref)


def test_nms_rotated_float64_is_double_precision(dev, oracle_lib):
    """float64 boxes (nms_rotated_cuda.cu:96 dispatches double): obb_nms_rotated_f64 orders by the double scores and decides
    every pair with the double-precision IoU (policy RotGeom64) -- the oracle's double scan (pinned to the reference's own
    header: golden riou_ref_dev_f64), index for index.  (a) a well-conditioned detector-like set; (b) a set where float32
    cannot work: coordinates around 3e7 (one float32 ulp = 2 px) with sub-pixel structure, near-identical boxes whose IoU
    differs from the threshold in the 9th digit, scores that are equal in float32 -- the float32 scan of the rounded boxes
    gives a DIFFERENT kept list there (asserted), the double path must still equal the double oracle; (c) the wrapper's
    small-box filter and empty / all-small inputs in float64."""
    from yolov5_obb_amd import nms_rotated_ext
    from yolov5_obb_amd.utils.nms_rotated import obb_nms
    g = torch.Generator().manual_seed(1)
    dets, scores = synth.s_clustered(6000, 120, 9)
    d64 = dets.double() + 1e-9 * torch.randn(dets.shape, dtype=torch.float64, generator=g)
    s64 = synth.tie_free(scores).double()
    s64[100] = s64[200] + 1e-12                                   # equal in float32, ordered in double
    for thr in (0.4, 0.1):
        got = nms_rotated_ext.nms_rotated(d64.to(dev), s64.to(dev), thr).cpu().numpy()
        assert np.array_equal(got, oracle.nms_rotated(d64.numpy(), s64.numpy(), thr, threads=8)), thr
    # (b)
    n = 5000
    base, sc = synth.s_clustered(n, 60, 13, extent=300.0)
    hard = base.double()
    hard[:, :2] += 3.0e7 + torch.rand(n, 2, dtype=torch.float64, generator=g)            # sub-ulp structure at 3e7
    hard[:, 2:4] *= 1.0 + 1e-9 * torch.randn(n, 2, dtype=torch.float64, generator=g)
    hs = synth.tie_free(sc).double() * (1.0 + 1e-13 * torch.arange(n, dtype=torch.float64))
    ref64 = oracle.nms_rotated(hard.numpy(), hs.numpy(), 0.4, threads=8)
    order = np.argsort(-hs.numpy(), kind="stable")
    ref32 = order[oracle.nms_rotated(hard.float().numpy()[order], np.arange(n, 0, -1, dtype=np.float32), 0.4, threads=8)]
    assert not np.array_equal(ref64, ref32)                        # the set discriminates: float32 decides differently here
    for _ in range(2):
        got = nms_rotated_ext.nms_rotated(hard.to(dev), hs.to(dev), 0.4).cpu().numpy()
        assert np.array_equal(got, ref64)
    # (c)
    small = d64.clone()
    pick = torch.randperm(len(small), generator=g)[:500]
    small[pick, 3] = 0.0005
    keep_mask = small[:, 2:4].min(1)[0] >= 0.001
    idx = torch.nonzero(keep_mask).squeeze(1).numpy()
    want = idx[oracle.nms_rotated(small[keep_mask].numpy(), s64[keep_mask].numpy(), 0.4, threads=8)]
    _, inds = obb_nms(small.to(dev), s64.to(dev), 0.4)
    assert inds.dtype == torch.int64 and np.array_equal(inds.cpu().numpy(), want)
    small[:, 3] = 0.0001
    assert len(obb_nms(small.to(dev), s64.to(dev), 0.4)[1]) == 0
    assert len(nms_rotated_ext.nms_rotated(d64[:0].to(dev), s64[:0].to(dev), 0.4)) == 0


def test_small_grid_fallback_gives_the_same_result(dev, oracle_lib):
    """obb_nms_set_max_grid(8): the grid the host layer retries with after a barrier abort (workgroups not co-resident);
    same kept list with 8 workgroups as with one per CU -- single list with the in-kernel index build, and the fused driver."""
    from yolov5_obb_amd import _lib, nms_rotated_ext
    from yolov5_obb_amd.utils.general import non_max_suppression_obb
    from oracle import pyref
    dets, scores = synth.s_clustered(40000, 2500, seed=5)
    scores = synth.tie_free(scores)
    ref = oracle.nms_rotated(dets.numpy(), scores.numpy(), 0.4, threads=8)
    pred = synth.s_pred(2, 4000, 15, seed=3)
    kw = dict(conf_thres=0.25, iou_thres=0.45, multi_label=True, max_det=300)
    want = pyref.non_max_suppression_obb(pred.clone(), **kw)
    # ... and a list that falls apart into six independent slabs: with 8 workgroups every one of them scatters ten tiles of its
    # block of positions (the one-tile form of k_slab_split's scatter is what the full grid runs)
    dets6, _ = synth.with_classes(dets, 6, 5)
    ref6 = oracle.nms_rotated(dets6.numpy(), scores.numpy(), 0.4, threads=8)
    L = _lib.lib()
    try:
        L.obb_nms_set_max_grid(8)
        got = nms_rotated_ext.nms_rotated(dets.to(dev), scores.to(dev), 0.4).cpu().numpy()
        got6 = nms_rotated_ext.nms_rotated(dets6.to(dev), scores.to(dev), 0.4).cpu().numpy()
        out = non_max_suppression_obb(pred.to(dev), **kw)
    finally:
        L.obb_nms_set_max_grid(0)
    assert np.array_equal(ref, got)
    assert np.array_equal(ref6, got6)
    assert np.array_equal(ref6, nms_rotated_ext.nms_rotated(dets6.to(dev), scores.to(dev), 0.4).cpu().numpy())      # (full grid: one tile)
    assert all(torch.equal(g.cpu(), w) for g, w in zip(out, want))


def test_nms_rotated_errors(dev):
    from yolov5_obb_amd import nms_rotated_ext
    from yolov5_obb_amd.utils.nms_rotated import obb_nms, poly_nms
    dets, scores = synth.s_uniform(10, 0)
    with pytest.raises(RuntimeError):
        nms_rotated_ext.nms_rotated(dets, scores, 0.5)            # CPU tensors: no CPU path in this build
    with pytest.raises(TypeError):
        obb_nms([1, 2, 3], scores, 0.5)
    with pytest.raises(NotImplementedError):
        poly_nms(torch.zeros(3, 9), 0.5)                          # reference: CPU poly_nms not implemented
    with pytest.raises(RuntimeError):
        nms_rotated_ext.nms_poly(torch.zeros(3, 9), 0.5)


@pytest.mark.parametrize("n,seed,thr", [(1, 0, 0.3), (300, 1, 0.1), (2500, 2, 0.3)])
def test_nms_poly(dev, oracle_lib, n, seed, thr):
    from yolov5_obb_amd import nms_rotated_ext
    from yolov5_obb_amd.utils.nms_rotated import poly_nms
    dets, scores = synth.s_clustered(n, max(1, n // 12), seed, extent=300.0)
    scores = synth.tie_free(scores)
    polys = torch.cat([synth.rbox_to_quad(dets), scores[:, None]], 1).contiguous()
    polys[::9, :8] = polys[::9, :8].reshape(-1, 4, 2).flip(1).reshape(-1, 8)   # some clockwise rings
    ref = oracle.nms_poly(polys.numpy(), thr)
    got = nms_rotated_ext.nms_poly(polys.to(dev), thr).cpu().numpy()
    assert np.array_equal(ref, got)
    kept, inds = poly_nms(polys.to(dev), thr)
    assert np.array_equal(inds.cpu().numpy(), ref) and torch.equal(kept.cpu(), polys[ref])


def test_nms_poly_30k(dev, oracle_lib):
    """the quad NMS at the reference's max_nms size (30,000 candidates, utils/general.py:794), the
    S-clustered quads bench.py times, against the oracle's restatement of poly_nms_cuda.cu:197-261 (14 M devPolyIoU
    evaluations on the host); twice."""
    from yolov5_obb_amd import nms_rotated_ext
    dq, sq = synth.s_clustered(30000, 300, seed=0)
    sq = synth.tie_free(sq)
    polys = torch.cat([synth.rbox_to_quad(dq), sq[:, None]], 1).contiguous()
    ref = oracle.nms_poly(polys.numpy(), 0.4)
    assert 300 < len(ref) < 3000
    for _ in range(2):
        got = nms_rotated_ext.nms_poly(polys.to(dev), 0.4).cpu().numpy()
        assert np.array_equal(ref, got)


@pytest.mark.parametrize("thr", [0.0, 0.05, 0.4, -0.1])
def test_nms_poly_skip_rule(dev, oracle_lib, thr):
    """The quad NMS skips a pair only when both boxes carry a finite bounding box (piou_device.h quad_cull_box: area large
    enough against the coordinate magnitude that the rounding noise of the reference's origin-based sum cannot reach the
    threshold).  A set that mixes every case -- tile coordinates, boxes shifted by 5,000 and by 70,000 (class-offset style:
    their noise is visible, nothing may be skipped), few-pixel boxes, zero-area quads (the reference's union == 0 rule
    makes two of them suppress each other at any distance), clockwise rings -- against the oracle, which clips every pair."""
    from yolov5_obb_amd import nms_rotated_ext
    d0, s0 = synth.s_clustered(1500, 60, seed=5, extent=1024.0)
    d1, s1 = synth.s_clustered(600, 30, seed=6, extent=900.0)
    d1[:, :2] += 5000.0
    d2, s2 = synth.s_clustered(600, 30, seed=7, extent=900.0)
    d2[:, :2] += 70000.0
    d3, s3 = synth.s_uniform(300, 8, extent=1024.0)
    d3[:, 2:4] = d3[:, 2:4].clamp(max=4.0)                                   # few-pixel boxes
    dets = torch.cat([d0, d1, d2, d3])
    scores = synth.tie_free(torch.cat([s0, s1, s2, s3]))
    quads = synth.rbox_to_quad(dets)
    quads[::40] = quads[::40, :2].repeat(1, 4)                               # zero-area quads (a point)
    quads[5::9] = quads[5::9].reshape(-1, 4, 2).flip(1).reshape(-1, 8)        # clockwise rings
    polys = torch.cat([quads, scores[:, None]], 1).contiguous()
    ref = oracle.nms_poly(polys.numpy(), thr)
    got = nms_rotated_ext.nms_poly(polys.to(dev), thr).cpu().numpy()
    assert np.array_equal(ref, got)


@pytest.mark.parametrize("extent,thr", [(4096.0, 0.1), (4096.0, 0.4), (30000.0, 0.3), (200.0, 0.0)])
def test_nms_poly_second_cone_rule_large_extents(dev, oracle_lib, extent, thr):
    """Round 6: the second proved cone rule inside the NMS kernels (QuadGeom::classify_quick: tier 1 on the extended cone, tier 2 on
    the plain cone + the pair check; QuadGeom::cheap_reject for budget-less quads).  On large extents rule B has no budget (its noise
    bound grows with the square of the coordinates) and the two cone rules decide almost every pair: 4,000 clustered + uniform quads
    incl. clockwise rings and integer coordinates against the oracle, which clips every pair; thr = 0 on a small extent (no rule B at
    thr <= 0: every quad is budget-less and carries the rule's words in its hot-loop record)."""
    from yolov5_obb_amd import nms_rotated_ext
    d0, s0 = synth.s_clustered(3000, 150, seed=31, extent=extent)
    d1, s1 = synth.s_uniform(1000, 32, extent=extent)
    dets = torch.cat([d0, d1])
    scores = synth.tie_free(torch.cat([s0, s1]))
    quads = synth.rbox_to_quad(dets)
    quads[3::11] = quads[3::11].reshape(-1, 4, 2).flip(1).reshape(-1, 8)
    quads[4::13] = quads[4::13].round()
    polys = torch.cat([quads, scores[:, None]], 1).contiguous()
    ref = oracle.nms_poly(polys.numpy(), thr)
    for _ in range(2):
        got = nms_rotated_ext.nms_poly(polys.to(dev), thr).cpu().numpy()
        assert np.array_equal(ref, got)
    assert 10 < len(ref) < 4000


def test_nms_poly_strict_equals_skip_100k(dev, tmp_path):
    """The skip rule of the quad NMS (csrc/piou_device.h) against the same library with the rule switched off
    (OBB_NMS_POLY_STRICT=1: every pair is clipped, as the reference does) at N = 100,000 quads in three layouts -- a 1024 px
    tile, the same shifted by 5,000 px, and 18 class offsets of 4096 px (utils/general.py:849-851) -- at two thresholds: the kept
    lists must be identical.  (The switch is read once per process: two child processes.)"""
    import subprocess
    import sys
    helper = os.path.join(os.path.dirname(os.path.abspath(__file__)), "helpers", "poly_strict_vs_skip.py")
    res = {}
    for strict in ("0", "1"):
        out = str(tmp_path / f"kept_{strict}.npz")
        env = dict(os.environ, OBB_NMS_POLY_STRICT=strict)
        subprocess.run([sys.executable, helper, out], check=True, env=env, timeout=900)
        res[strict] = np.load(out)
    assert sorted(res["0"].files) == sorted(res["1"].files) and len(res["0"].files) == 6
    for k in res["0"].files:
        assert np.array_equal(res["0"][k], res["1"][k]), k
        assert len(res["0"][k]) > 100


def test_nms_poly_fresh_seed_every_rule_against_no_rule(dev, tmp_path):
    """Rule B of the quad NMS is a searched bound, used only inside the envelope the search covered
    (csrc/piou_device.h: |coordinate| <= 70,000, bounding box <= 600 x 600).  Every run of the suite extends the search: 30,000
    quads of six adversarial families ON BOTH SIDES of the envelope, drawn from a FRESH seed (printed; pass OBB_TEST_SEED to
    repeat one), through the library with both rules (default), with the proved rule only (OBB_NMS_POLY_STRICT=1) and with no
    rule at all (=2: every pair clipped, as the reference does), at three thresholds: identical kept lists."""
    import subprocess
    import sys
    import time
    seed = int(os.environ.get("OBB_TEST_SEED", int(time.time()) % 1000000007))
    print(f"fresh seed of this run: {seed}")
    helper = os.path.join(os.path.dirname(os.path.abspath(__file__)), "helpers", "poly_strict_vs_skip.py")
    res = {}
    for strict in ("0", "1", "2"):
        out = str(tmp_path / f"fresh_{strict}.npz")
        subprocess.run([sys.executable, helper, out, "fresh", str(seed)], check=True, env=dict(os.environ, OBB_NMS_POLY_STRICT=strict), timeout=900)
        res[strict] = np.load(out)
    for k in res["2"].files:
        assert len(res["2"][k]) > 1000, (seed, k)
        assert np.array_equal(res["0"][k], res["2"][k]), f"seed {seed}, {k}: rules A + B differ from the clip of every pair"
        assert np.array_equal(res["1"][k], res["2"][k]), f"seed {seed}, {k}: rule A differs from the clip of every pair"


def test_ops_rbox_overlaps_device_tensors(dev, oracle_lib):
    """ops.rbox_overlaps -> obb_rbox_overlaps_f32 (the device-pointer form of the devkit's overlaps_kernel,
    poly_overlaps_kernel.cu:280-353): same matrix as the host-pointer `_overlaps`, bit for bit, and as the oracle.  The one
    thing no build can pin is the reference's `cos(float)` (CUDA's cosf, <= 2 ulp); kernel and oracle both take the correctly
    rounded float (double cos / sin rounded once: round 4; with each side's own cosf the matrices differed by up to 4.8e-5),
    so they agree bit for bit unless the two double libms differ in a last bit that flips a float rounding (~2^-29 per angle).
    Shapes incl. empty and one row."""
    from yolov5_obb_amd import ops
    from yolov5_obb_amd.DOTA_devkit.poly_nms_gpu import poly_overlaps
    a, _ = synth.s_uniform(700, 17, extent=300.0)
    b, _ = synth.s_uniform(333, 18, extent=300.0)
    got = ops.rbox_overlaps(a.to(dev), b.to(dev)).cpu().numpy()
    ref = oracle.devkit_overlaps(a.numpy(), b.numpy())
    assert got.shape == (700, 333) and got.dtype == np.float32
    err = np.abs(got - ref)
    same = got.view(np.uint32) == ref.view(np.uint32)
    assert same.mean() >= 0.999 and err.max() <= 1e-5, (same.mean(), err.max())          # north_star: 1e-5
    assert np.array_equal(got, poly_overlaps(a.numpy(), b.numpy()))          # the two entry points share the kernel
    assert ops.rbox_overlaps(a[:1].to(dev), b.to(dev)).shape == (1, 333)
    assert ops.rbox_overlaps(a[:0].to(dev), b.to(dev)).shape == (0, 333)
    assert (got > 0).sum() > 1000 and got.max() <= 1.0 + 1e-6


def test_rotated_iou_pairs_bit_exact(dev, oracle_lib):
    from yolov5_obb_amd import ops
    a, _ = synth.s_uniform(200000, 1, extent=120.0)
    b, _ = synth.s_uniform(200000, 2, extent=120.0)
    a[:1000] = b[:1000]
    a[1000:2000, 4] = 0; b[1000:2000, 4] = 0
    a[2000:3000, :4] = a[2000:3000, :4].round(); b[2000:3000, :4] = b[2000:3000, :4].round()
    a[2000:3000, 4] = 0; b[2000:3000, 4] = 0
    got = ops.rotated_iou_pairs(a.to(dev), b.to(dev)).cpu().numpy()
    ref = oracle.riou_pairs(a.numpy(), b.numpy())
    bad = got.view(np.uint32) != ref.view(np.uint32)
    # double-precision cos/sin come from ocml on the GPU and glibc on the host: a last-ulp difference in the
    # double result can flip the float rounding of cos/sin once in ~1e8 evaluations; allow <= 2 such pairs
    assert bad.sum() <= 2, (bad.sum(), np.abs(got - ref).max())
    assert np.abs(got - ref).max() <= 1e-5
    assert (ref > 0).mean() > 0.2


def test_rotated_iou_matrix(dev, oracle_lib):
    from yolov5_obb_amd import ops
    a, _ = synth.s_uniform(300, 3, extent=200.0)
    b, _ = synth.s_uniform(257, 4, extent=200.0)
    got = ops.rotated_iou_matrix(a.to(dev), b.to(dev)).cpu().numpy()
    ref = oracle.riou_matrix(a.numpy(), b.numpy())
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))


def test_quad_iou_matrix_bit_exact(dev, oracle_lib):
    from yolov5_obb_amd import ops
    a, _ = synth.s_uniform(260, 5, extent=150.0)
    b, _ = synth.s_uniform(190, 6, extent=150.0)
    qa, qb = synth.rbox_to_quad(a), synth.rbox_to_quad(b)
    qb[::5] = qb[::5].reshape(-1, 4, 2).flip(1).reshape(-1, 8)
    qa[3] = qa[3, :2].repeat(4)        # degenerate point
    qb[7] = qb[7, :2].repeat(4)
    got = ops.quad_iou_matrix(qa.to(dev), qb.to(dev)).cpu().numpy()
    ref = oracle.piou_matrix(qa.numpy(), qb.numpy())
    same = (got.view(np.uint32) == ref.view(np.uint32)) | (np.isnan(got) & np.isnan(ref))
    assert same.all(), (np.count_nonzero(~same), np.nanmax(np.abs(got - ref)))
    assert got[3, 7] == 1.0            # both degenerate: union == 0 -> (0+1)/(0+1)  (poly_nms_cuda.cu:136-137)


@pytest.mark.parametrize("extent,na,nb,seed", [(1024.0, 300, 700, 21), (4096.0, 130, 520, 22), (60.0, 257, 65, 23), (30000.0, 64, 1000, 24)])
def test_quad_iou_matrix_both_cone_rules_bit_exact(dev, oracle_lib, extent, na, nb, seed):
    """k_quad_strip writes an exact +0 where either cone rule of piou_device.h fires (the column quad counter-clockwise of the row
    quad as seen from the origin, or -- round 6, quad_cone2_skip -- clockwise of it) and clips the rest: every entry equals the
    oracle's full clip bit for bit, on extents where nine pairs in ten are such zeros, across the 256-column blocks' edges, with
    reversed rings, a quad around the origin, quads whose edge lines pass through the origin's neighbourhood and integer grids."""
    from yolov5_obb_amd import ops
    a, _ = synth.s_uniform(na, seed, extent=extent)
    b, _ = synth.s_uniform(nb, seed + 100, extent=extent)
    qa, qb = synth.rbox_to_quad(a), synth.rbox_to_quad(b)
    qb[::5] = qb[::5].reshape(-1, 4, 2).flip(1).reshape(-1, 8)
    qa[::7] = qa[::7].reshape(-1, 4, 2).flip(1).reshape(-1, 8)
    qa[1::9] = qa[1::9].round(); qb[2::9] = qb[2::9].round()
    qa[5] = torch.tensor([-3.0, -2.0, 40.0, -2.0, 40.0, 30.0, -3.0, 30.0])                       # around the origin: no cone
    qb[6] = torch.tensor([10.0, 10.0, 500.0, 500.5, 499.0, 502.0, 9.0, 11.5])                    # edge lines through the origin's neighbourhood
    qa[8] = torch.tensor([0.0, 50.0, 0.0, 90.0, -20.0, 90.0, -20.0, 50.0])                       # a vertex on the y axis, x <= 0
    got = ops.quad_iou_matrix(qa.to(dev), qb.to(dev)).cpu().numpy()
    ref = oracle.piou_matrix(qa.numpy(), qb.numpy())
    same = (got.view(np.uint32) == ref.view(np.uint32)) | (np.isnan(got) & np.isnan(ref))
    assert same.all(), (np.count_nonzero(~same), np.nanmax(np.abs(got - ref)))
    # the devkit flavour (rboxes in, RotBox2Poly on the device) on the same boxes
    gd = ops.rbox_overlaps(a.to(dev), b.to(dev)).cpu().numpy()
    rd = oracle.devkit_overlaps(a.numpy(), b.numpy())
    assert (gd.view(np.uint32) == rd.view(np.uint32)).mean() >= 0.999 and np.abs(gd - rd).max() <= 1e-5


def test_devkit_overlaps_and_poly_gpu_nms(dev, oracle_lib):
    from yolov5_obb_amd.DOTA_devkit.poly_nms_gpu import poly_overlaps
    from yolov5_obb_amd.DOTA_devkit.poly_nms_gpu.nms_wrapper import poly_nms_gpu
    from yolov5_obb_amd.DOTA_devkit.poly_nms_gpu.poly_nms import poly_gpu_nms
    a, _ = synth.s_uniform(150, 7, extent=150.0)
    b, _ = synth.s_uniform(70, 8, extent=150.0)
    got = poly_overlaps(a.numpy(), b.numpy())
    ref = oracle.devkit_overlaps(a.numpy(), b.numpy())
    assert got.shape == (150, 70) and got.dtype == np.float32
    assert (got.view(np.uint32) == ref.view(np.uint32)).mean() >= 0.999 and np.abs(got - ref).max() <= 1e-5
    dets, scores = synth.s_clustered(1200, 80, 9, extent=300.0)
    scores = synth.tie_free(scores)
    polys = torch.cat([synth.rbox_to_quad(dets), scores[:, None]], 1).numpy()
    order = polys[:, 8].argsort()[::-1]            # poly_nms.pyx:18-19
    ref_keep = order[oracle.devkit_poly_nms(np.ascontiguousarray(polys[order]), 0.3)]
    got_keep = poly_gpu_nms(polys, 0.3)
    assert isinstance(got_keep, list) and np.array_equal(np.asarray(got_keep), ref_keep)
    assert poly_nms_gpu(polys[:0], 0.3) == []


def test_heavy_duplication_is_repeatable(dev, oracle_lib):
    """Few objects with thousands of near-duplicates each: the chunk's conflict graph has far more edges than fit in
    LDS (resolve runs on the global edge list), several steps with a cross phase each.  Every repetition must give the
    oracle's kept list -- this is the configuration that exposes inter-workgroup visibility mistakes."""
    import oracle
    for n, k, seed, thr in ((12000, 12, 5, 0.3), (30000, 40, 6, 0.5), (6000, 3, 7, 0.45)):
        d, s = synth.s_clustered(n, k, seed)
        s = synth.tie_free(s)
        ref = oracle.nms_rotated(d.numpy(), s.numpy(), thr)
        dd, ss = d.to(dev), s.to(dev)
        from yolov5_obb_amd import nms_rotated_ext
        for rep in range(6):
            got = nms_rotated_ext.nms_rotated(dd, ss, thr).cpu().numpy()
            assert np.array_equal(got, ref), (n, k, rep, len(got), len(ref))


regime_100k = synth.regime_100k


@pytest.mark.parametrize("regime", ["clustered_k300", "clustered_k300_raw", "clustered_k300_18cls", "clustered_k3000", "clustered_k3000_18cls",
                                    "uniform", "uniform_18cls"])
def test_full_size_100k_exact(dev, oracle_lib, regime):
    """BASELINE.json configs[3] size: the kept list of the HIP NMS at N = 100,000, iou 0.4, equals the oracle's -- same
    indices, same order -- on every regime bench.py times; four device runs each (the kernels' work distribution depends
    on timing, the result must not; the first call of a size class takes the persistent kernel, the later ones the path the
    library chooses from what that call reported -- the phase kernels of csrc/nms_mk.h for K=3000 and S-uniform).  The oracle needs 2.3 s for S-clustered and about four minutes for S-uniform on one
    thread; its inner loop is split over the host's cores here (same result for any thread count)."""
    import os
    dets, scores = regime_100k(regime)
    thr = 0.4
    ref = oracle.nms_rotated(dets.numpy(), scores.numpy(), thr, threads=min(os.cpu_count() or 1, 64))
    d, s = dets.to(dev), scores.to(dev)
    from yolov5_obb_amd import nms_rotated_ext
    for rep in range(4):
        got = nms_rotated_ext.nms_rotated(d, s, thr).cpu().numpy()
        assert len(got) == len(ref), (regime, rep, len(got), len(ref))
        assert np.array_equal(ref, got), (regime, rep)


def test_full_size_100k_properties(dev, oracle_lib):
    """BASELINE.json configs[3] size (100k candidates, S-clustered, iou 0.4), checked through size-independent properties
    next to the exact comparison above: (a) the kept indices come in descending score order, (b) kept boxes do not suppress
    each other (pairwise IoU <= thr, from the bit-exact IoU matrix kernel), (c) idempotence: NMS of the kept set keeps all
    of it, (d) completeness on a sample: every sampled dropped box has a kept box with a higher score and IoU > thr.
    (a)-(d) together characterise the greedy result."""
    from yolov5_obb_amd import nms_rotated_ext, ops
    thr = 0.4
    dets, scores = synth.s_clustered(100000, 300, seed=0)
    scores = synth.tie_free(scores)
    d, s = dets.to(dev), scores.to(dev)
    keep = nms_rotated_ext.nms_rotated(d, s, thr)
    k = keep.cpu().numpy()
    assert 100 < len(k) < 5000 and len(np.unique(k)) == len(k)
    ks = scores.numpy()[k]
    assert np.all(ks[:-1] > ks[1:])                                            # (a)
    kd = d[keep]
    m = ops.rotated_iou_matrix(kd, kd).cpu().numpy()                           # rows = higher score
    iu = np.triu_indices(len(k), 1)
    assert (m[iu] <= thr).all()                                                # (b)
    again = nms_rotated_ext.nms_rotated(kd, s[keep], thr).cpu().numpy()
    assert np.array_equal(again, np.arange(len(k)))                            # (c)
    rng = np.random.RandomState(0)
    dropped = np.setdiff1d(np.arange(len(scores)), k)
    sample = rng.choice(dropped, 4000, replace=False)
    iou = ops.rotated_iou_matrix(kd, d[torch.from_numpy(sample).to(dev)]).cpu().numpy()     # (kept, sample)
    higher = ks[:, None] > scores.numpy()[sample][None, :]
    assert ((iou > thr) & higher).any(0).all()                                 # (d)


def _slab_case(variant, n=30000):
    """Lists that fall apart into groups that cannot overlap (the callers' cls * 4096 offsets, utils/general.py:849-851)."""
    g = torch.Generator().manual_seed(11)
    flags = 0
    if variant in ("cls18", "cls2", "cls90", "ties", "dropped"):
        nc = {"cls18": 18, "cls2": 2, "cls90": 90}.get(variant, 7)
        dets, scores = synth.s_clustered(n, 200, seed=3)
        dets, _ = synth.with_classes(dets, nc, 3)
        if variant == "ties":
            scores = scores.half().float()                                      # heavy ties: ascending-index rule across slabs
        if variant == "dropped":
            pick = torch.randperm(n, generator=g)[:1500]
            dets[pick, 3] = 0.0005
    elif variant == "x_only":                                                   # offsets on x alone, uneven groups
        dets, scores = synth.s_clustered(n, 150, seed=4)
        cls = (torch.rand(n, generator=g) ** 2 * 5).long()
        dets[:, 0] += cls.float() * 3000.0
    elif variant == "y_only":                                                   # groups apart on y only: one slab on x, the list stays whole
        dets, scores = synth.s_clustered(n, 150, seed=4)
        dets[:, 1] += torch.randint(0, 6, (n,), generator=g).float() * 5000.0
    elif variant == "big_slab":                                                 # one group above the slab limit: the call stays one list
        dets, scores = synth.s_clustered(n + 30000, 300, seed=5)
        dets[:8000, 0] += 9000.0
    elif variant == "needle":                                                   # one ill-conditioned box switches the decomposition off
        dets, scores = synth.s_clustered(n, 200, seed=6)
        dets, _ = synth.with_classes(dets, 9, 6)
        dets[17, 2] = 0.004
    elif variant == "touching":                                                 # groups whose circles just touch / just do not
        dets, scores = synth.s_uniform(n, 8, extent=600.0)
        dets[:, 0] = (dets[:, 0] % 150.0) + torch.randint(0, 12, (n,), generator=g).float() * 236.0
    else:
        raise KeyError(variant)
    if variant != "ties":
        scores = synth.tie_free(scores)
    if variant == "dropped":
        from yolov5_obb_amd import _lib
        flags = _lib.OBB_NMS_DROP_SMALL
    return dets, scores, flags


@pytest.mark.parametrize("variant", ["cls18", "cls2", "cls90", "ties", "dropped", "x_only", "y_only", "big_slab", "needle", "touching"])
def test_independent_slabs_same_result(dev, oracle_lib, variant):
    """Slab mode of the single-list kernel (csrc/grid.h "independent slabs", nms_core.h slab_setup / slab_merge): the list
    is re-laid out slab by slab inside the kernel and run as concurrent segments; the kept list must be the oracle's,
    index for index, in global score order -- for 2 / 18 / more than 64 groups, ties, boxes dropped by the small-box
    filter, uneven groups, and for the inputs that must NOT be decomposed (no gap on x, a group above the limit, an
    ill-conditioned box).  Three runs each: the work distribution depends on timing, the result must not."""
    import os
    from yolov5_obb_amd import nms_rotated_ext
    dets, scores, flags = _slab_case(variant)
    thr = 0.4
    if variant == "dropped":
        keep_mask = dets[:, 2:4].min(1)[0] >= 0.001
        idx = torch.nonzero(keep_mask).squeeze(1).numpy()
        ref = idx[oracle.nms_rotated(dets[keep_mask].numpy(), scores[keep_mask].numpy(), thr, threads=min(os.cpu_count() or 1, 32))]
    else:
        ref = oracle.nms_rotated(dets.numpy(), scores.numpy(), thr, threads=min(os.cpu_count() or 1, 32))
    d, s = dets.to(dev), scores.to(dev)
    for rep in range(3):
        got = nms_rotated_ext._run_rotated(d, s, thr, flags=flags).cpu().numpy()
        assert len(got) == len(ref), (variant, rep, len(got), len(ref))
        assert np.array_equal(ref, got), (variant, rep)


def _desc_order(scores):
    """The documented order of the sort in front of the NMS: descending score, NaN first (torch's order), -0 == +0, ties by
    ascending original index -- restated with numpy on the score bits."""
    s = np.asarray(scores, dtype=np.float32)
    u = s.view(np.uint32).copy()
    u[s == 0] = 0
    k = np.where(u & 0x80000000, ~u, u | np.uint32(0x80000000)).astype(np.uint32)
    k[np.isnan(s)] = 0xFFFFFFFF
    return np.lexsort((np.arange(len(s)), ~k))


def _lattice(n):
    """n small boxes that cannot touch each other: the NMS keeps every one, its output IS the sort order."""
    i = torch.arange(n)
    return torch.stack([(i % 512).float() * 8, (i // 512).float() * 8, torch.full((n,), 2.0), torch.full((n,), 2.0), torch.zeros(n)], 1)


@pytest.mark.parametrize("case", ["random", "equal", "ascending", "descending", "interleaved", "ties", "specials"])
@pytest.mark.parametrize("n", [1, 2, 511, 512, 513, 1000, 70001, 131072, 131073])
def test_own_sort_order(dev, n, case):
    """The three-launch sort (csrc/psrs_sort.h; n > 131072: the radix sort of csrc/segsort.h) through the NMS entry point, on
    boxes that do not interact.  `interleaved`: every run of 512 consecutive elements holds 31 top scores -- the input that
    drives one bucket of the regular-sampling partition to its bound (the rank-counting merge instead of the in-LDS network)."""
    if n > 1000 and case in ("equal", "ascending", "ties") and n != 131072:
        pytest.skip("covered at the other sizes")
    g = torch.Generator().manual_seed(n)
    s = torch.rand(n, generator=g)
    if case == "equal":
        s = torch.full((n,), 0.5)
    elif case == "ascending":
        s = torch.sort(s)[0]
    elif case == "descending":
        s = torch.sort(s, descending=True)[0]
    elif case == "interleaved":
        s = s * 0.5
        top = (torch.arange(n) % 512) < 31
        s[top] = 0.9 + 0.1 * torch.rand(int(top.sum()), generator=g)
    elif case == "ties":
        s = (s * 50).round() / 50
    elif case == "specials":
        s = s - 0.5
        idx = torch.randperm(n, generator=g)[: max(1, n // 7)]
        vals = torch.tensor([float("nan"), 0.0, -0.0, float("inf"), float("-inf"), 1e-45, -1e-45])
        s[idx] = vals[torch.arange(len(idx)) % len(vals)]
    dets = _lattice(n)
    got = _gpu_keep(dets, s, 0.5, dev)
    assert np.array_equal(got, _desc_order(s.numpy()))


def test_persistent_kernel_under_load_from_another_stream(dev, oracle_lib):
    """The persistent NMS kernel needs all its workgroups resident at once (team barriers).  val.py inside a training process
    shares the device with whatever else is queued: here a second stream keeps every CU busy with long GEMMs (>= 5 ms each,
    LDS-heavy tiles) while the 100k-box NMS and the bs16 fused step run on the current stream.  The results must equal the
    oracle's; the latency is bounded (the workgroups that are resident wait for the others at bounded spin barriers: the worst
    case is an abort + one retry on an 8-workgroup grid, never a hang) and is reported."""
    import time
    import warnings
    from oracle import pyref
    from yolov5_obb_amd import nms_rotated_ext
    from yolov5_obb_amd.utils import general
    d, s = synth.regime_100k("clustered_k300")
    ref = oracle.nms_rotated(d.numpy(), s.numpy(), 0.4)
    dg, sg = d.to(dev), s.to(dev)
    pred = synth.s_pred(16, 64512, 16, seed=1000, n_obj=120, fg_frac=0.03, device=dev, dtype=torch.float16)
    kw = dict(conf_thres=0.25, iou_thres=0.45, multi_label=True, max_det=1500)
    ref_step = pyref.non_max_suppression_obb(pred.cpu().clone(), **kw)
    for _ in range(2):                                            # warm both paths (workspaces, the candidate hint)
        nms_rotated_ext.nms_rotated(dg, sg, 0.4)
        general.non_max_suppression_obb(pred, **kw)
    torch.cuda.synchronize()

    from yolov5_obb_amd import _lib
    retries0 = _lib.abort_retries()

    def timed(fn):
        # the call's own latency: nms_rotated returns when ITS stream has delivered the kept count (the binding polls a pinned
        # word), the current stream is drained for good measure -- not the device: the GEMM queue of the side stream is not the call's
        t0 = time.perf_counter(); out = fn(); torch.cuda.current_stream().synchronize(); return out, (time.perf_counter() - t0) * 1e3
    _, quiet_nms = timed(lambda: nms_rotated_ext.nms_rotated(dg, sg, 0.4))
    _, quiet_step = timed(lambda: general.non_max_suppression_obb(pred, **kw))
    a = torch.randn(8192, 8192, device=dev)
    side = torch.cuda.Stream(device=dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(side):
        e0.record(); c = a @ a; e1.record()
    torch.cuda.synchronize()
    gemm_ms = e0.elapsed_time(e1)
    lat_nms, lat_step = [], []
    for rep in range(3):
        with torch.cuda.stream(side):                             # ~40 GEMMs in flight behind each other on the other stream
            for _ in range(40):
                c = a @ a
        time.sleep(0.002)                                         # let the first GEMM take the CUs
        k, ms = timed(lambda: nms_rotated_ext.nms_rotated(dg, sg, 0.4))
        lat_nms.append(ms)
        torch.cuda.synchronize()
        assert np.array_equal(k.cpu().numpy(), ref)
        with torch.cuda.stream(side):
            for _ in range(40):
                c = a @ a
        time.sleep(0.002)
        t0 = time.perf_counter()
        out = general.non_max_suppression_obb(pred, **kw)         # returns after its own count read-back: the call's latency
        lat_step.append((time.perf_counter() - t0) * 1e3)
        torch.cuda.synchronize()
        assert len(out) == len(ref_step)
        for g_, r_ in zip(out, ref_step):
            assert torch.equal(g_[:, 5].cpu(), r_[:, 5]) and np.array_equal(synth.canon_rows(g_), synth.canon_rows(r_))
    del c
    busy = 40 * gemm_ms
    retries = _lib.abort_retries() - retries0
    warnings.warn(UserWarning(f"NMS under load: GEMM {gemm_ms:.2f} ms each; quiet nms100k {quiet_nms:.2f} ms / step {quiet_step:.2f} ms; "
                              f"beside 40 queued GEMMs: fused step latency {[round(x, 2) for x in lat_step]} ms, "
                              f"nms100k call latency {[round(x, 1) for x in lat_nms]} ms (the GEMM queue alone: {busy:.0f} ms); "
                              f"calls repeated on the 8-workgroup grid after a barrier time-out: {retries}"))
    assert gemm_ms >= 5.0
    assert max(lat_step) < busy + 2000.0 and max(lat_nms) < busy + 2000.0
