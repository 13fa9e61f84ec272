#!/usr/bin/env python3
"""profiles/<tag>_pmc_fetch.md + <tag>_pmc_write.md (tools/rocpd_pmc.py output of the two --pmc passes over
tools/prof_kernels.py) -> the per-launch HBM byte counts bench.py reports as `traffic`.
usage: python tools/pmc_json.py profiles/r2_pmc_fetch.md profiles/r2_pmc_write.md [profiles/r2_hotpath_kernel_stats.md] > profiles/r2_pmc.json

bytes per launch = 2 * FETCH_SIZE[KB] * 1024 + WRITE_SIZE[KB] * 1024: MI355X_MICROARCH.md -- FETCH_SIZE tallies 128-byte
requests at 64 bytes on gfx950; the 256 MiB calibration copy at the start of prof_kernels.py must read FETCH 131072 KB,
WRITE 262144 KB (checked here)."""
import json
import re
import sys


def per_dispatch(path):
    out = {}
    for line in open(path):
        m = re.match(r"- `(.+?)` (\w+): (.*)", line)
        if m:
            out[m.group(1)] = [float(v) for v in m.group(3).split(", ")]
    return out


def table(path):
    out = {}
    for line in open(path):
        m = re.match(r"\| `(.+?)` \| (\w+) \| (\d+) \| ([\d.e+]+) \| ([\d.e+]+) \| ([\d.e+]+) \| ([\d.e+]+) \|", line)
        if m:
            out[m.group(1)] = dict(n=int(m.group(3)), per=float(m.group(5)), mn=float(m.group(6)), mx=float(m.group(7)))
    return out


def find(d, key):
    for k, v in d.items():
        if key in k:
            return v
    raise KeyError(key)


def main():
    fpath, wpath = sys.argv[1], sys.argv[2]
    f, w = per_dispatch(fpath), per_dispatch(wpath)
    ft, wt = table(fpath), table(wpath)
    cal_f, cal_w = find(ft, "__amd_rocclr_copyBuffer")["mx"], find(wt, "__amd_rocclr_copyBuffer")["mx"]
    assert abs(cal_f - 131072) < 200 and abs(cal_w - 262144) < 200, (cal_f, cal_w)

    def mean(v):
        return sum(v) / len(v)

    def bytes_(fk, wk):
        return int(round(2 * fk * 1024 + wk * 1024))
    out = {"_source": f"{fpath}, {wpath} (rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE -- python tools/prof_kernels.py, "
                      "separate passes); bytes per launch = 2 * FETCH_SIZE[KB] * 1024 + WRITE_SIZE[KB] * 1024 (MI355X_MICROARCH.md: "
                      "FETCH_SIZE tallies 128-byte requests at 64 bytes on gfx950; confirmed by the 256 MiB calibration copy of the "
                      f"same run: FETCH {cal_f:.0f} KB, WRITE {cal_w:.0f} KB)"}
    for name, key in (("k_decode", "obb::k_decode<"), ("k_loss_bwd_dense", "obb::k_loss_bwd_dense<"),
                      ("k_loss_dense_fwd", "obb::k_loss_dense_fwd<")):
        fk, wk = mean(find(f, key)), mean(find(w, key))
        out[name] = bytes_(fk, wk)
        out[name + "_raw_kb"] = {"FETCH_SIZE": round(fk, 1), "WRITE_SIZE": round(wk, 1)}
    # prof_kernels.py: REPS bs16 steps (the multi-segment kernel), then REPS 100k calls of S-clustered K=300 and REPS of the same
    # with 18 class offsets (the single-list kernel with the indexed cross phase)
    # the bs16 step's NMS kernel: round 4's small-segment kernel (the first call of the shape may still run the persistent one)
    try:
        fb, wb = find(f, "obb::k_nms_small<obb::RotGeom"), find(w, "obb::k_nms_small<obb::RotGeom")      # (round 5: <RotGeom, SmallGather>)
        out["k_nms_bs16_kernel"] = ("obb::k_nms_small<obb::RotGeom, obb::SmallGather, obb::SmallSelfSort>" if any("SmallSelfSort" in k and "k_nms_small" in k for k in f) else
                                    "obb::k_nms_small<obb::RotGeom, obb::SmallGather>" if any("SmallGather" in k and "k_nms_small" in k for k in f) else "obb::k_nms_small<obb::RotGeom>")
    except KeyError:
        fb, wb = find(f, "obb::k_nms_persist<obb::RotGeom, false>"), find(w, "obb::k_nms_persist<obb::RotGeom, false>")
        out["k_nms_bs16_kernel"] = "obb::k_nms_persist<obb::RotGeom, false>"
    out["k_nms_persist_bs16"] = bytes_(mean(fb), mean(wb))             # (key kept: bench.py reads it)
    out["k_nms_persist_bs16_raw_kb"] = {"FETCH_SIZE": round(mean(fb), 1), "WRITE_SIZE": round(mean(wb), 1)}
    fn, wn = find(f, "obb::k_nms_persist<obb::RotGeom, true>"), find(w, "obb::k_nms_persist<obb::RotGeom, true>")
    h = len(fn) // 2
    for name, sl in (("k_nms_persist_100k_clustered_k300_raw", slice(0, h)), ("k_nms_persist_100k_clustered_k300_18cls", slice(h, None))):
        fk, wk = mean(fn[sl]), mean(wn[sl])
        out[name] = bytes_(fk, wk)
        out[name + "_raw_kb"] = {"FETCH_SIZE": round(fk, 1), "WRITE_SIZE": round(wk, 1)}
    out["k_nms_persist_100k"] = out["k_nms_persist_100k_clustered_k300_raw"]
    try:                                               # all levels in one launch (obb_detect_decode_levels, Detect.forward's default)
        fd, wd = find(f, "obb::k_detect_decode_levels<"), find(w, "obb::k_detect_decode_levels<")
        out["k_detect_decode"] = bytes_(mean(fd), mean(wd))
        out["k_detect_decode_raw_kb"] = {"FETCH_SIZE": round(mean(fd), 1), "WRITE_SIZE": round(mean(wd), 1)}
    except KeyError:                                   # one launch per level
        fd, wd = find(f, "obb::k_detect_decode<"), find(w, "obb::k_detect_decode<")
        lv_f = [mean(fd[i::3]) for i in range(3)]
        lv_w = [mean(wd[i::3]) for i in range(3)]
        out["k_detect_decode"] = sum(bytes_(a, b) for a, b in zip(lv_f, lv_w))
        out["k_detect_decode_raw_kb"] = {"FETCH_SIZE": [round(v, 1) for v in lv_f], "WRITE_SIZE": [round(v, 1) for v in lv_w]}
    if len(sys.argv) > 3:                              # kernel-trace summary of the same driver: avg duration of k_loss_bwd_dense
        for line in open(sys.argv[3]):
            if "obb::k_loss_bwd_dense<" in line:
                out["k_loss_bwd_dense_ms"] = round(float(line.split("|")[4]) * 1e-3, 5)
    json.dump(out, sys.stdout, indent=1)
    print()


main()
