"""CPU: the native reader / writer of Task1 files (csrc/textio.hip, host code only) and the column-wise detection reader
against their line-by-line twins, which follow the reference's text handling statement by statement (DOTA_devkit/ResultMerge_multi_process.py:186-233,
dota_evaluation_task1.py:152-160).  Bit-equal doubles, identical output lines."""
import numpy as np
import pytest

from tests.golden import gen_golden as gg



@pytest.mark.parametrize("cfg", list(gg.MERGE_CASES.values()) + [(50, 30, 9, True)])
def test_result_table_equals_line_by_line_parse(tmp_path, cfg):
    from yolov5_obb_amd.DOTA_devkit import ResultMerge_multi_process as RM
    src = tmp_path / "Task1_plane.txt"
    src.write_text("\n".join(gg.merge_input_lines(*cfg)) + "\n")
    boxes = RM.parse_result_file(str(src))
    table = RM.parse_result_table(str(src))
    names, codes, dets = table.names, table.codes, table.dets
    assert names == list(boxes)
    for g, nm in enumerate(names):
        a, b = np.array(boxes[nm]), dets[codes == g]
        assert a.shape == b.shape and np.array_equal(a.view(np.uint64), b.view(np.uint64))
    want = [RM.format_result_line(names[c], dets[i].tolist()) for i, c in enumerate(codes)]
    assert RM.format_result_rows([names[c] for c in codes], dets) == want
    assert table.format_rows(np.arange(len(dets))).decode().splitlines() == want


def test_unusual_lines_send_the_native_reader_back_to_the_line_by_line_path(tmp_path):
    from yolov5_obb_amd.DOTA_devkit import ResultMerge_multi_process as RM
    good = "P1__1__0___824 0.5 1 2 3 4 5 6 7 8"
    for bad in ("P1__1__0___824 0.5 1 2 3 4 5 6 7", "P1__1__0___824  0.5 1 2 3 4 5 6 7 8", "P1_1_0_824 0.5 1 2 3 4 5 6 7 8",
                "P1__1__0___824 0.5 1 2 3 4 5 6 7 1_0", "P1__1__0___824 0.5 1 2 3 4 5 6 7 nan", ""):
        f = tmp_path / "t.txt"
        f.write_text(good + "\n" + bad + "\n" + good + "\n")
        assert RM.parse_result_table(str(f)) is None
    f.write_text(good + "\r\n  " + good.replace("P1__1__", "Q__0.5__") + "\t\n" + good)       # CRLF, padding, no final newline
    t = RM.parse_result_table(str(f))
    assert t.names == ["P1", "Q"] and t.codes.tolist() == [0, 1, 0]
    assert np.array_equal(t.dets[1], np.array([(1 + 0) / 0.5, (2 + 824) / 0.5, 6.0, 1656.0, 10.0, 1660.0, 14.0, 1664.0, 0.5]))


def test_row_formatter_follows_python_round_and_str():
    from yolov5_obb_amd.DOTA_devkit import ResultMerge_multi_process as RM
    v = np.array([[0.15, 0.25, 0.35, 1.0, 2.5, 1234.05, -0.04, 0.05, 0.125], [99.95, 0.049999, 0.005, 7.0, 3.14159, 1e-9, 2.675, 0.5, 1.0],
                  [0.45, 0.55, 0.65, 0.75, 0.85, 0.95, 1.05, 1.15, 0.995], [10.0, 100.0, 0.0, 5.55, 6.65, 7.75, 8.85, 9.95, 0.3]])
    assert RM.format_result_rows(['a'] * 4, v) == [RM.format_result_line('a', r.tolist()) for r in v]
    rng = np.random.RandomState(0)
    w = np.round(rng.rand(20000, 9) * 3000, 2)
    w[:, 8] = np.round(rng.rand(20000), 5)
    assert RM.format_result_rows(['b'] * 20000, w) == [RM.format_result_line('b', r.tolist()) for r in w]


def test_detection_reader_equals_line_by_line(tmp_path):
    pytest.importorskip("pandas")
    from yolov5_obb_amd.DOTA_devkit import dota_evaluation_task1 as EV
    _, det = gg.eval_inputs(20, 30, 4)
    f = tmp_path / "Task1_plane.txt"
    f.write_text("\n".join(det["plane"]) + "\n")
    ids, conf, bb = EV.read_detections(str(f))
    split = [x.strip().split(' ') for x in det["plane"]]
    assert ids == [x[0] for x in split]
    assert np.array_equal(conf, np.array([float(x[1]) for x in split]))
    assert np.array_equal(bb, np.array([[float(z) for z in x[2:]] for x in split]))


def test_native_reader_never_disagrees_with_the_line_by_line_parse(tmp_path):
    """Fuzz (hypothesis): for arbitrary mixes of well-formed and odd lines the native reader either declines (None) or
    returns exactly what the statement-by-statement Python parse returns -- it never invents a different reading."""
    hyp = pytest.importorskip("hypothesis")
    from hypothesis import given, settings, strategies as st
    from yolov5_obb_amd.DOTA_devkit import ResultMerge_multi_process as RM

    good = st.one_of(st.floats(min_value=-5000, max_value=5000, allow_nan=False).map(repr),
                     st.floats(min_value=0, max_value=4000).map(lambda v: f"{v:.2f}"),
                     st.sampled_from(["12", "12.5", "-3.25", "1e2", "1E-2", "+4", "007", "0.1", ".5", "5.", "3.141592653589793", "4.9e-324"]))
    odd = st.sampled_from(["1_0", "nan", "inf", "0x10", "", "1e", "--1", "1e400", "123456789012345678"])
    num = st.one_of(*([good] * 80 + [odd]))               # mostly well-formed fields, so that whole files get accepted
    rate = st.sampled_from(["1", "0.5", "1.0", "2", "0.25", "+1", "1.", ".5"] * 3 + ["1+1", ""])
    stem = st.sampled_from(["P0001", "P0002", "P_1", "a__b", "x", "P0001_", "_P", "图"])
    tile = st.builds(lambda s, r, x, y, sep: f"{s}__{r}__{x}{sep}{y}", stem, rate, st.sampled_from(["0", "824", "12", "1648"] * 4 + [""]),
                     st.sampled_from(["0", "1648", "824"] * 5 + [""]), st.sampled_from(["___"] * 30 + ["__", "____"]))
    sep = st.sampled_from([" "] * 300 + ["  ", "\t"])
    line = st.builds(lambda t, nums, seps, pad: pad[0] + "".join(a + b for a, b in zip([t] + nums, seps + [""])) + pad[1], tile,
                     st.integers(0, 19).flatmap(lambda k: st.lists(num, min_size=9 if k else 8, max_size=9 if k else 10)),
                     st.lists(sep, min_size=10, max_size=10),
                     st.sampled_from([("", ""), (" ", ""), ("", " \r"), ("\t", "  ")]))
    f = tmp_path / "t.txt"

    @settings(max_examples=1500, deadline=None)
    @given(st.lists(line, min_size=1, max_size=3), st.booleans())
    def check(lines, final_newline):
        f.write_bytes(("\n".join(lines) + ("\n" if final_newline else "")).encode())
        t = RM.parse_result_table(str(f))
        seen[t is not None] += 1
        if t is None:
            return
        boxes = RM.parse_result_file(str(f))              # must not raise when the native reader accepted the file
        assert t.names == list(boxes)
        for g, nm in enumerate(t.names):
            a, b = np.array(boxes[nm], dtype=np.float64), t.dets[t.codes == g]
            assert a.shape == b.shape and np.array_equal(a.view(np.uint64), b.view(np.uint64)), (lines, nm)
    seen = {True: 0, False: 0}
    check()
    assert seen[True] >= 10 and seen[False] >= 20, seen     # both outcomes were exercised (about 3 % of the files are accepted)


def test_fast_paths_of_the_number_conversions_agree_with_python(tmp_path):
    """csrc/textio.hip converts the usual numbers without strtod / snprintf (exact integer arithmetic): 400k values across all
    magnitudes, binary fractions that are exact rounding ties, signed zeros, subnormals, values next to the fast path's limits
    -- against Python's own round()/str() and float()."""
    from yolov5_obb_amd.DOTA_devkit import ResultMerge_multi_process as RM
    rng = np.random.RandomState(7)
    # (str(round(v, d)) is the d-decimal string as long as that string has at most 15 significant digits: |v| < 10^(15-d))
    mags = 10.0 ** rng.uniform(-14, 12.99, 200000) * rng.choice([-1.0, 1.0], 200000)
    ties = (rng.randint(-4000000, 4000000, 150000) + rng.choice([0.5, 0.25, 0.75, 0.125, 0.375, 0.625, 0.875, 0.0625, 0.03125], 150000)) / \
        rng.choice([1.0, 2.0, 4.0, 8.0, 16.0, 64.0, 1024.0], 150000)
    special = np.array([0.0, -0.0, 5e-324, -5e-324, 2.2250738585072014e-308, 0.05, 0.15, 0.25, 0.35, 0.45, 0.005, 0.015, 0.025, 0.995,
                        0.9999999, 9.95, 99.95, 999.95, 9999999999999.994, 9999999999999.996, 8796093022207.995, 4398046511103.125,
                        0.049999999999999996, 0.05000000000000001, 1e-3, 4.35, 2.675, 1.005, 1234567890123.455, 0.994999999999999996])
    vals = np.concatenate([mags, ties, special, -special])
    vals = np.concatenate([vals, np.zeros((-len(vals)) % 9)]).reshape(-1, 9)
    got = RM.format_result_rows(['n'] * len(vals), vals)
    assert got == [RM.format_result_line('n', r.tolist()) for r in vals]
    # ---- decimal -> double: literals of every shape the fast path takes or must decline
    lits = [f"{v:.{d}f}" for v, d in zip(10.0 ** rng.uniform(-6, 13, 60000) * rng.choice([-1.0, 1.0], 60000), rng.randint(0, 8, 60000))]
    lits += [repr(float(v)) for v in 10.0 ** rng.uniform(-5, 6, 20000)]                       # 17 significant digits: the general path
    lits += [f"{rng.randint(0, 10 ** 15)}" for _ in range(5000)] + [f"0.{rng.randint(0, 10 ** 15):022d}" for _ in range(5000)]
    lits += ["999999999999999", "1000000000000000", "0.0000000000000000000001", "0.00000000000000000000001", "-0.0", "+0", "007.50", ".5",
             "5.", "123456789012345.6", "12345678901234.56", "9007199254740993", "0.1", "0.2", "0.3", "179769313486231570000", "1e22", "1e23",
             "4.9e-324", "8.5", "-8.25", "000000000000000000001", "100000000000000000000.5"]
    lits = [x for x in lits if len(x) <= 60]
    rows = [lits[i:i + 9] for i in range(0, len(lits) - 8, 9)]
    f = tmp_path / "Task1_x.txt"
    f.write_text("\n".join("P0__1__0___0 " + " ".join(r) for r in rows) + "\n")
    t = RM.parse_result_table(str(f))
    assert t is not None and len(t.dets) == len(rows)
    want = np.array([[(float(x) + 0.0) / 1.0 for x in r[1:]] + [float(r[0])] for r in rows])   # poly2origpoly with x = y = 0, rate 1
    assert np.array_equal(t.dets.view(np.uint64), want.view(np.uint64))


def _gt_arrays_from_records(EV, annopath, imagenames, classname):
    gts, gt_off, difficult = [], [0], []
    for imagename in imagenames:
        objs = [o for o in EV.parse_gt(annopath.format(imagename)) if o['name'] == classname]
        gts.extend(o['bbox'] for o in objs)
        difficult.extend(bool(o['difficult']) for o in objs)
        gt_off.append(len(gts))
    return np.array(gts, dtype=np.float64).reshape(-1, 8), np.array(difficult, dtype=np.bool_), np.array(gt_off, dtype=np.int64)


def test_native_ground_truth_reader_equals_parse_gt(tmp_path):
    from yolov5_obb_amd.DOTA_devkit import dota_evaluation_task1 as EV
    gt, det = gg.eval_inputs(25, 20, 6)
    detpath, annopath, imagesetfile = gg.eval_write(str(tmp_path), gt, det)
    names = [x.strip() for x in open(imagesetfile)]
    # odd but legal files: short lines and blank lines are skipped, CRLF, no difficult flag, names in another script, a class
    # that is a prefix of another one, padding, a file without records
    (tmp_path / "odd").mkdir()
    odd = str(tmp_path / "odd" / "{:s}.txt")
    files = {
        "a": "imagesource:GoogleEarth\ngsd:0.1\n1 2 3 4 5 6 7 8 plane 0\n1.5 2.5 3 4 5 6 7 8 plane 1\n\n9 8 7 6 5 4 3 2 planes 0\n",
        "b": "1 2 3 4 5 6 7 8 plane\r\n  10 20 30 40 50 60 70 80 plane 2  \r\n1 2 3 4 5 6 7 8 船 0\r\n",
        "c": "",
        "d": "0.1 0.2 0.3 0.4 0.5 0.6 0.7 0.8 plane 0",
        "e": "1 2 3 4 5 6 7 8 ship 1\n1e2 2E1 +3 .4 5. 006 7 8 plane 0\n",
    }
    for k, v in files.items():
        with open(odd.format(k), "w", newline="") as f:
            f.write(v)
    for ap, nm, classes in ((annopath, names, ("plane", "ship", "nothing")), (odd, list(files) + ["a"], ("plane", "planes", "船", "ship"))):
        for cls in classes:
            got = EV.load_gt(ap, nm, cls)
            assert got is not None
            want = _gt_arrays_from_records(EV, ap, nm, cls)
            assert np.array_equal(got[0].view(np.uint64), want[0].view(np.uint64)) and np.array_equal(got[1], want[1]) and np.array_equal(got[2], want[2])
    assert len(EV.load_gt(odd, list(files), "plane")[0]) == 6
    # declined (the record-by-record path then behaves like the reference, including its exceptions)
    for bad in ("1 2 3 4 5 6 7 8 plane 0 extra\n", "1 2 3 4 5 6 7 x plane 0\n", "1 2 3 4 5 6 7 8  plane 0\n", "1 2 3 4 5 6 7 8 plane 1_0\n",
                "1 2 3 4 5 6 7 8 plane +1\n", "1 2 3 4 5 6 7 8 plane 0\rnext line\n", "1 2 3 4 5 6 7 8 plane nan\n", "\x1c1 2 3 4 5 6 7 8 plane 0\n",
                "1 2 3 4 5 6 7 8 plane 0 \n", "1 2 3 4 5 6 7 8 plane 0　\n"):
        with open(odd.format("z"), "w", newline="") as f:
            f.write(bad)
        assert EV.load_gt(odd, ["a", "z"], "plane") is None, bad


def test_native_detection_reader_declines_what_it_should(tmp_path):
    from yolov5_obb_amd.DOTA_devkit import dota_evaluation_task1 as EV
    f = tmp_path / "Task1_plane.txt"
    good = "P0001 0.5 1 2 3 4 5 6 7 8"
    f.write_text(good + "\n  图片__1 1e-3 1.5 2.5 3 4 5 6 7 8  \r\n" + good)
    ids, conf, bb = EV._read_detections_native(str(f))
    assert ids == ["P0001", "图片__1", "P0001"] and conf.tolist() == [0.5, 0.001, 0.5] and bb[1].tolist() == [1.5, 2.5, 3, 4, 5, 6, 7, 8]
    for bad in (good + " 9", good[:-2], good.replace(" 0.5", "  0.5"), good.replace("0.5", "nan"), good.replace("0.5", "1_0"), "",
                good + "\rP2 0.5 1 2 3 4 5 6 7 8", "\x1d" + good, good + " "):
        f.write_text(good + "\n" + bad + "\n", newline="")
        assert EV._read_detections_native(str(f)) is None, bad


def test_voc_eval_host_logic_with_a_cpu_stand_in_for_the_device_call(tmp_path, monkeypatch, oracle_lib):
    """voc_eval end to end on the CPU: the device call (best_gt) is replaced by the oracle's per-detection routine, everything
    around it -- the native readers, the index arithmetic, the TP/FP pass, AP -- is the shipped code.  Against the restated
    reference evaluation (oracle/pyref.py) on the same files."""
    from oracle import pyref
    from yolov5_obb_amd.DOTA_devkit import dota_evaluation_task1 as EV
    gt, det = gg.eval_inputs(30, 25, 5)
    detpath, annopath, imagesetfile = gg.eval_write(str(tmp_path), gt, det)

    def best_gt_cpu(dets8, det_img, gts8, gt_off):
        ov = np.empty(len(dets8)); jm = np.empty(len(dets8), dtype=np.int32)
        for d in range(len(dets8)):
            g = gts8[gt_off[det_img[d]]:gt_off[det_img[d] + 1]]
            o, j = pyref.task1_best_gt(np.asarray(dets8[d], dtype=float), np.asarray(g, dtype=float))
            ov[d], jm[d] = o, (-1 if j is None else j)
        return ov, jm
    monkeypatch.setattr(EV, "best_gt", best_gt_cpu)
    names = [x.strip() for x in open(imagesetfile)]
    parsed = {k: EV.parse_gt(annopath.format(k)) for k in names}
    for cls in ("plane", "ship"):
        if cls not in det:
            continue
        for use07 in (True, False):
            rec, prec, ap = EV.voc_eval(detpath, annopath, imagesetfile, cls, 0.5, use07)
            r2, p2, a2 = pyref.task1_voc_eval(parsed, names, det[cls], cls, 0.5, use07)
            assert np.array_equal(rec, r2) and np.array_equal(prec, p2) and ap == a2
