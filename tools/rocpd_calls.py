#!/usr/bin/env python3
"""Per-CALL sums from a rocprofv3 rocpd database of tools/mk_trace.py (one regime, several nms_rotated calls): the dispatches are cut
into calls at every k_ps_local_scores (the first kernel of a call), the first `skip` calls are left out (path choice and step hints
settle), and for the rest: per kernel name the dispatches per call, the mean duration, the time per call, and -- when the database
was collected with --pmc -- the counter sums per call.  A 256 MiB calibration copy in front of the calls (MK_CALIB=1) is reported
separately.        python tools/rocpd_calls.py <results.db> [skip] -> JSON"""
import json
import sqlite3
import sys


def main():
    db = sys.argv[1]
    skip = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    cur = sqlite3.connect(db).cursor()
    kcols = [r[1] for r in cur.execute("pragma table_info('kernels')")]
    s_col = "start" if "start" in kcols else "start_timestamp"
    e_col = "end" if "end" in kcols else "end_timestamp"
    rows = cur.execute(f"select dispatch_id, name, {s_col}, {e_col} from kernels order by {s_col}").fetchall()
    vals = {}
    try:
        for did, cn, v in cur.execute("select dispatch_id, counter_name, counter_value from pmc_events"):
            d = vals.setdefault(did, {})
            d[cn] = d.get(cn, 0.0) + float(v)
            d["_rows_" + cn] = d.get("_rows_" + cn, 0) + 1
    except sqlite3.OperationalError:
        pass
    short = lambda n: n.split("(")[0].replace("void ", "").replace("obb::", "")
    calib = [vals.get(did, {}) for did, name, s, e in rows if "copyBuffer" in name or "elementwise_kernel" in name and False]
    starts = [i for i, r in enumerate(rows) if "k_ps_local_scores" in r[1]]
    calls = []
    for a, b in zip(starts, starts[1:] + [len(rows)]):
        seg = [r for r in rows[a:b] if "obb::" in r[1]]
        calls.append(seg)
    calls = calls[skip:]
    out = {"calls": len(calls), "kernels": {}, "per_call": {}}
    if not calls:
        print(json.dumps(out)); return
    nc = len(calls)
    tot_us = 0.0
    csum = {}
    for seg in calls:
        for did, name, s, e in seg:
            k = out["kernels"].setdefault(short(name), {"dispatches": 0, "us": 0.0, "counters": {}})
            k["dispatches"] += 1
            k["us"] += (e - s) / 1e3
            tot_us += (e - s) / 1e3
            dv = vals.get(did, {})
            for cn, v in dv.items():
                if cn.startswith("_rows_"):
                    continue
                if cn == "GRBM_GUI_ACTIVE":           # summed over the XCDs by rocprofv3 (one row), or one row per XCD: cycles of ONE XCD
                    nr = dv.get("_rows_" + cn, 1)
                    v = v / (nr if nr > 1 else 8)
                k["counters"][cn] = k["counters"].get(cn, 0.0) + v
                csum[cn] = csum.get(cn, 0.0) + v
    for k in out["kernels"].values():
        k["avg_us"] = round(k["us"] / max(1, k["dispatches"]), 3)
        k["dispatches_per_call"] = round(k["dispatches"] / nc, 2)
        k["us_per_call"] = round(k["us"] / nc, 3)
        k["counters_per_call"] = {cn: v / nc for cn, v in k.pop("counters").items()}
        del k["us"], k["dispatches"]
    out["per_call"] = {"kernel_us": round(tot_us / nc, 2), "span_us": round(sum((seg[-1][3] - seg[0][2]) / 1e3 for seg in calls) / nc, 2),
                       "dispatches": round(sum(len(s) for s in calls) / nc, 2), "counters": {cn: v / nc for cn, v in csum.items()}}
    cal = {}
    for did, name, s, e in rows:
        if "copyBuffer" in name or "copy_kernel" in name.lower():
            for cn, v in vals.get(did, {}).items():
                if not cn.startswith("_rows_"):
                    cal[cn] = max(cal.get(cn, 0.0), v)
    out["calibration_copy_max"] = cal
    print(json.dumps(out))


if __name__ == "__main__":
    main()
