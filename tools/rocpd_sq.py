#!/usr/bin/env python3
"""SQ counter tables from rocprofv3 rocpd databases collected with --kernel-trace --pmc <SQ counters> over tools/prof_sq.py
(one database per pass; the passes are joined on (kernel, launch index)).

usage: python tools/rocpd_sq.py <title> <reps> <db1> [<db2> ...]   -> markdown

Per dispatch of every obb:: kernel: duration, the raw counters, and
  valu_frac      = 4 * SQ_ACTIVE_INST_VALU / (SIMDS * GRBM_GUI_ACTIVE)    (the counter is in quad-cycles, summed over the waves of the device;
                   SIMDS = 256 CUs x 4: the share of the device's VALU issue slots that held a VALU instruction)
  valu_issue     = 4 * SQ_INSTS_VALU / (SIMDS * GRBM_GUI_ACTIVE)          (instructions x 4 cycles: the same from the issue side)
  wait_frac      = SQ_WAIT_ANY / SQ_WAVE_CYCLES,   stall_frac = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES,   busy_frac = SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES
  waves_per_simd = SQ_WAVE_CYCLES * 4 / (SIMDS * GRBM_GUI_ACTIVE)         (average resident waves per SIMD over the kernel)
  lds_conflict   = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
GRBM_GUI_ACTIVE is summed over the XCDs by rocprofv3 (8 on MI355X): it is divided by XCDS before use."""
import sqlite3
import sys

SIMDS = 256 * 4
XCDS = 8
# tools/prof_sq.py launch order: workload of the i-th group of `reps` dispatches of a kernel
GROUPS = {
    "k_nms_persist<obb::RotGeom, true>": ["clustered_k300", "clustered_k300_18cls", "clustered_k3000", "uniform"],
    "k_slab_split<obb::RotGeom>": ["clustered_k300", "clustered_k300_18cls", "clustered_k3000", "uniform"],
    "k_prep_rot": ["clustered_k300", "clustered_k300_18cls", "clustered_k3000", "uniform"],
    "k_ps_local_scores": ["clustered_k300", "clustered_k300_18cls", "clustered_k3000", "uniform"],
    "k_ps_split": ["clustered_k300", "clustered_k300_18cls", "clustered_k3000", "uniform", "nms_poly_30k"],
    "k_ps_bucket": ["clustered_k300", "clustered_k300_18cls", "clustered_k3000", "uniform", "nms_poly_30k"],
}


def short(n):
    n = n.replace("void ", "").replace("obb::", "", 1)
    return n.split("(")[0][:70]


def load(db):
    cur = sqlite3.connect(db).cursor()
    kern = {}
    for did, name, s, e in cur.execute("select dispatch_id, name, start, end from kernels order by start"):
        kern[did] = (name, s, e)
    vals, inst = {}, {}
    for did, cn, v in cur.execute("select dispatch_id, counter_name, counter_value from pmc_events"):
        vals.setdefault(did, {})
        vals[did][cn] = vals[did].get(cn, 0.0) + float(v)
        inst[(did, cn)] = inst.get((did, cn), 0) + 1
    for (did, cn), k in inst.items():                 # GRBM_GUI_ACTIVE: the mean over the instances rocprofv3 reports (one row if it
        if cn == "GRBM_GUI_ACTIVE":                   # has summed them already: then the XCDS division below applies)
            vals[did]["_gui_rows"] = k
    per = {}
    for did in sorted(kern, key=lambda d: kern[d][1]):
        name, s, e = kern[did]
        if "obb::" not in name:
            continue
        per.setdefault(short(name), []).append(((e - s) / 1e3, vals.get(did, {})))
    return per


def main():
    title, reps = sys.argv[1], int(sys.argv[2])
    passes = [load(p) for p in sys.argv[3:]]
    print(f"# {title}\n")
    print(__doc__.split("usage:")[1].split("\n", 1)[1])
    kernels = []
    for p in passes:
        for k in p:
            if k not in kernels:
                kernels.append(k)
    for k in kernels:
        n = max(len(p.get(k, [])) for p in passes)
        rows = []
        for i in range(n):
            dur, c = [], {}
            for p in passes:
                if i < len(p.get(k, [])):
                    dur.append(p[k][i][0])
                    for cn, v in p[k][i][1].items():
                        if cn == "_gui_rows":
                            continue
                        if cn == "GRBM_GUI_ACTIVE":
                            rows_ = p[k][i][1].get("_gui_rows", 1)
                            c.setdefault("_gui", []).append(v / (rows_ if rows_ > 1 else XCDS))
                        else:
                            c[cn] = v
            gl = c.pop("_gui", [])
            gui = sum(gl) / max(1, len(gl))
            rows.append((i, sum(dur) / max(1, len(dur)), gui, c))
        names = sorted({cn for _, _, _, c in rows for cn in c})
        if not names:
            continue
        print(f"\n## `{k}`\n")
        grp = None
        for key, g in GROUPS.items():
            if k.startswith(key):
                grp = g
        print("| # | workload | us (profiled) | cycles/XCD | valu_frac | valu_issue | waves/SIMD | busy | stall | wait | lds_conflict | " + " | ".join(names) + " |")
        print("|---:|---|---:|---:|---:|---:|---:|---:|---:|---:|---:|" + "---:|" * len(names))
        for i, dur, gui, c in rows:
            wl = grp[i // reps] if grp and i // reps < len(grp) else ""
            g = lambda n: c.get(n, float("nan"))
            cyc = gui if gui > 0 else float("nan")
            wc = g("SQ_WAVE_CYCLES")
            f = lambda x: "" if x != x else f"{x:.3f}"
            print(f"| {i} | {wl} | {dur:.1f} | {cyc:.0f} | {f(4 * g('SQ_ACTIVE_INST_VALU') / (SIMDS * cyc))} | {f(4 * g('SQ_INSTS_VALU') / (SIMDS * cyc))} | "
                  f"{f(4 * wc / (SIMDS * cyc))} | {f(g('SQ_ACTIVE_INST_ANY') / wc)} | {f(g('SQ_WAIT_INST_ANY') / wc)} | {f(g('SQ_WAIT_ANY') / wc)} | "
                  f"{f(g('SQ_LDS_BANK_CONFLICT') / g('SQ_LDS_IDX_ACTIVE') if g('SQ_LDS_IDX_ACTIVE') == g('SQ_LDS_IDX_ACTIVE') and g('SQ_LDS_IDX_ACTIVE') > 0 else float('nan'))} | "
                  + " | ".join(f"{c.get(n, float('nan')):.6g}" for n in names) + " |")


if __name__ == "__main__":
    main()
