"""CPU: the PRODUCT's device geometry headers (yolov5_obb_amd/csrc/*_device.h) compiled with g++ and compared
bit-for-bit with the oracle on millions of seeded pairs, including the conservative rejects ('cull violations')."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build_and_run(src, exe, args, tmp_path):
    out = tmp_path / exe
    cmd = ["g++", "-O2", "-std=c++17", "-ffp-contract=off", f"-I{ROOT}/yolov5_obb_amd/csrc", f"{ROOT}/tests/native/{src}",
           f"{ROOT}/oracle/liboracle.so", f"-Wl,-rpath,{ROOT}/oracle", "-o", str(out), "-lm"]
    subprocess.run(cmd, check=True)
    r = subprocess.run([str(out)] + args, capture_output=True, text=True)
    return r.returncode, r.stdout


def test_rotated_iou_device_code_bit_exact(oracle_lib, tmp_path):
    rc, out = _build_and_run("host_check_riou.cpp", "hc_riou", ["1500000", "42"], tmp_path)
    assert rc == 0, out
    assert "mismatches=0 cull_violations=0 ub_violations=0" in out, out


def test_rotated_iou_double_device_code_bit_exact(oracle_lib, tmp_path):
    """csrc/riou64_device.h (the float64 NMS policy RotGeom64) against the oracle's double flavour, bit for bit."""
    rc, out = _build_and_run("host_check_riou64.cpp", "hc_riou64", ["600000", "44"], tmp_path)
    assert rc == 0, out
    assert "mismatches64=0" in out, out


def test_quad_iou_device_code_bit_exact(oracle_lib, tmp_path):
    rc, out = _build_and_run("host_check_piou.cpp", "hc_piou", ["400000", "43"], tmp_path)
    assert rc == 0, out
    assert "mismatches=0" in out and "mismatches64=0" in out, out


def test_quad_skip_rule_noise_bound(tmp_path):
    """piou_device.h quad_skip_record / quad_skip_pair: the rounding noise of the reference's origin-based quad intersection on
    bounding-box-disjoint pairs (ten adversarial families, |coord| 8 .. 70000, plus a greedy ascent that looks for inputs
    whose roundings line up) stays far below the bound the NMS skip rule is built on, no skipped pair is a hit or has
    overlapping bounding boxes, and the record's fp16 box is rounded outward."""
    out = tmp_path / "hc_quadcull"
    subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", f"-I{ROOT}/yolov5_obb_amd/csrc",
                    f"{ROOT}/tests/native/host_check_quadcull.cpp", "-o", str(out), "-lm"], check=True)
    r = subprocess.run([str(out), "4000000", "21", "20000"], capture_output=True, text=True)
    assert r.returncode == 0 and " wrong=0" in r.stdout and "fp16_rounding_errors=0" in r.stdout, r.stdout
    assert "cone_wrong=0" in r.stdout and "cone_skipped=0 " not in r.stdout, r.stdout      # the exact rule fired and never on a non-zero sum
    vals = dict(kv.split("=") for line in r.stdout.splitlines() if "=" in line and not line.startswith("family") for kv in line.split())
    bound = float(vals["bound_units"])
    assert float(vals["worst_noise_units"]) * 64 <= bound, r.stdout          # random pairs: measured < 8 units
    assert float(vals["worst_after_climb_units"]) * 32 <= bound, r.stdout    # after the ascent: measured < 20 units
    assert int(vals["culled"]) > 1000000, r.stdout


@pytest.mark.parametrize("fma", [False, True])
def test_quad_cone_rule_gives_exactly_zero(tmp_path, fma):
    """The PROVED skip rule of the quad IoU (piou_device.h quad_cone_skip): whenever the second quad's cone
    lies counter-clockwise of the first's, every one of the 16 terms of the reference's sum is exactly zero and the IoU is
    +0 -- checked on pairs generated to sit on the rule's edges (smallest resolvable gaps, spans up to pi, vertices at the minimum
    distance, coordinates 2 .. 1e7, bow ties, clockwise rings), in a build without and in a build WITH FMA contraction
    (nvcc's default for the reference's .cu files).
    Round 6: the SECOND proved rule (quad_cone2_skip / quad_cone2_nofuzzy: the first quad counter-clockwise of the second; tier 1
    on the extended-edge cone, tier 2 on the plain cone plus the pair check) in the build without contraction -- this project's
    arithmetic contract, the one the rule is stated for -- on the same pairs and on four families of its own: rectangles with
    the gap set around the smallest one the rule accepts for the pair's M / r, edges of one quad lying on an edge line of the other
    (0 .. 3 ulps off, 1e-9 .. 1e-3 off, and on lines next to an axis where clip 2's sign values land within +-2e-8: the
    extrapolated crossing), integer grids."""
    out = tmp_path / ("hc_cone_fma" if fma else "hc_cone")
    flags = ["-march=native", "-ffp-contract=fast"] if fma else ["-ffp-contract=off"]
    subprocess.run(["g++", "-O2", "-std=c++17", *flags, f"-I{ROOT}/yolov5_obb_amd/csrc",
                    f"{ROOT}/tests/native/host_check_quadcone.cpp", "-o", str(out), "-lm"], check=True)
    r = subprocess.run([str(out), "3000000", "5", "0" if fma else "1"], capture_output=True, text=True)
    assert r.returncode == 0 and " wrong=0" in r.stdout and " wrong2=0" in r.stdout, r.stdout + r.stderr
    vals = dict(kv.split("=") for kv in r.stdout.split())
    assert int(vals["fired"]) > 1500000 and int(vals["near_edge"]) > 500000, r.stdout
    if not fma:
        assert int(vals["fired2"]) > 1000000 and int(vals["tier1"]) > 300000 and int(vals["tier2"]) > 300000, r.stdout
        assert int(vals["at_edge2"]) > 200000 and min(int(v) for v in vals["family2_fired"].split(",")) > 30000, r.stdout


def test_fast_iou_interval_contains_the_reference_value(oracle_lib, tmp_path):
    """rbox_quick_bounds and rbox_fast_iou_bounds (the register-only filters in front of the exact clip): whenever one vouches for a pair, the
    oracle's IoU lies inside the interval -- detector-like distributions incl. nearly parallel, thin, class-offset,
    large, abutting and few-pixel boxes.  (The guard against corners that sit within rounding distance of the other
    box's boundary is what keeps the reference's own fragile cases out.)"""
    rc, out = _build_and_run("host_check_fastiou.cpp", "hc_fastiou", ["6000000", "11"], tmp_path)
    assert rc == 0, out
    assert "\nviolations=0" in out and "quick_bounds_violations=0" in out, out


def test_spatial_index_never_skips_a_pair_the_circle_test_keeps(tmp_path):
    """grid.h (the spatial index of the NMS cross phase), built and queried on the CPU with the kernels' own arithmetic:
    for sampled rows, every non-brute box that RotGeom::cheap_reject does not reject is among the visited candidates --
    uniform / clustered / class-offset / unit-square / mixed-size / outlier + degenerate / huge-coordinate sets."""
    out = tmp_path / "hc_grid"
    subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", f"-I{ROOT}/yolov5_obb_amd/csrc",
                    f"{ROOT}/tests/native/host_check_grid.cpp", "-o", str(out), "-lm"], check=True)
    for n, seed, fine in ((20000, 1, 0), (100000, 2, 0), (100000, 3, 1), (20000, 4, 2)):      # fine: cells of side 2 R_L / 2^fine
        r = subprocess.run([str(out), str(n), str(seed), str(fine)], capture_output=True, text=True)
        assert r.returncode == 0 and "\nviolations=0" in r.stdout, r.stdout
