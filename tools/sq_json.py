#!/usr/bin/env python3
"""profiles/<round>_sq.md (tools/rocpd_sq.py) -> profiles/<round>_sq.json: per (kernel, workload) the means over its dispatches of the
derived SQ figures (first dispatch of a group left out when the group has three or more: its GRBM_GUI_ACTIVE includes the gap in front
of it).  bench.py copies `valu_frac` / `wait_frac` of the 100k regimes into the `roofline` object.
usage: python tools/sq_json.py profiles/r5_sq.md > profiles/r5_sq.json"""
import json
import sys


def main():
    path = sys.argv[1]
    out = {"_source": f"{path} (rocprofv3 --kernel-trace --pmc, three passes over tools/prof_sq.py; formulas in the file's header)"}
    sec = None
    hdr = None
    rows = {}
    for line in open(path):
        if line.startswith("## `"):
            sec = line.split("`")[1].replace("obb::", "")
            hdr = None
            continue
        if sec and line.startswith("| # |"):
            hdr = [c.strip() for c in line.strip().strip("|").split("|")]
            continue
        if sec and hdr and line.startswith("|") and not line.startswith("|---"):
            c = [x.strip() for x in line.strip().strip("|").split("|")]
            if len(c) != len(hdr):
                continue
            r = dict(zip(hdr, c))
            rows.setdefault((sec, r["workload"]), []).append(r)
    short = {"k_nms_persist<RotGeom, true>": "k_nms_persist_100k", "k_nms_small<RotGeom>": "k_nms_small_bs16", "k_nms_small<RotGeom, SmallGather>": "k_nms_small_bs16", "k_nms_small<RotGeom, SmallGather, SmallSelfSort>": "k_nms_small_bs16", "k_nms_small<RotGeom, SmallGather, SmallFromSort>": "k_nms_small_bs16", "k_quad_strip<true>": "k_quad_strip",
             "k_decode<__half>": "k_decode", "k_sort_prep_lds": "k_sort_prep_lds", "k_gather_out": "k_gather_out",
             "k_nms_persist<QuadGeom, false>": "k_nms_persist_quad_30k", "k_ps_local_scores": "k_ps_local_scores", "k_ps_split": "k_ps_split",
             "k_ps_bucket": "k_ps_bucket", "k_prep_rot": "k_prep_rot", "k_slab_split<RotGeom>": "k_slab_split"}
    for (sec, wl), rs in rows.items():
        key = short.get(sec)
        if key is None:
            continue
        if len(rs) >= 3:
            rs = rs[1:]

        def mean(col):
            v = [float(r[col]) for r in rs if r.get(col) not in (None, "", "nan")]
            return round(sum(v) / len(v), 4) if v else None
        ent = {"dispatches": len(rs), "us_profiled": mean("us (profiled)"), "valu_frac": mean("valu_frac"), "valu_issue_frac": mean("valu_issue"),
               "waves_per_simd": mean("waves/SIMD"), "busy_frac": mean("busy"), "stall_frac": mean("stall"), "wait_frac": mean("wait"),
               "lds_conflict_frac": mean("lds_conflict")}
        out[key + ("_" + wl if wl else "")] = ent
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
