#!/usr/bin/env python3
"""gpurun_out/<tag>/nms100k_<regime>_<pass>.json (tools/r6_nms_prof.sh) -> profiles/r6_nms100k_kernel_stats.md, profiles/r6_sq.md,
profiles/r6_sq.json and the nms_100k_call_<regime> keys of profiles/r6_pmc.json (HBM bytes of the WHOLE call: every kernel between
k_ps_local_scores and k_finalize; bytes = 2 * FETCH_SIZE[KB] * 1024 + WRITE_SIZE[KB] * 1024 as MI355X_MICROARCH.md prescribes for
gfx950, checked on the 256 MiB calibration copy of the same run: FETCH 131072 KB, WRITE 262144 KB).
usage: python tools/r6_collect.py gpurun_out/<tag>"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REGIMES = ["clustered_k300_raw", "clustered_k300_18cls", "clustered_k3000", "clustered_k3000_18cls", "uniform"]
SIMDS = 256 * 4


def load(o, r, p):
    try:
        return json.load(open(os.path.join(o, f"nms100k_{r}_{p}.json")))
    except Exception:
        return None


def main():
    o = sys.argv[1]
    prof = os.path.join(ROOT, "profiles")
    ks = ["# Rotated NMS of one list, N = 100,000, per regime: the kernels of a call (round 6)\n",
          "`rocprofv3 --kernel-trace -- python tools/mk_trace.py <regime> 8` per regime (tools/r6_nms_prof.sh), on the path the un-profiled library chooses for the regime (pinned with OBB_NMS_MK: the profiler's per-dispatch overhead would change the choice); "
          "means over the last four calls (tools/rocpd_calls.py cuts the dispatches into calls at k_ps_local_scores).  Durations under the "
          "profiler's serialisation: the un-profiled call times are bench.py's `nms_100k`.\n"]
    sq_md = ["# SQ counters of the single-list rotated NMS at N = 100,000, per regime and kernel (round 6)\n",
             "`rocprofv3 --kernel-trace --pmc <counters> -- python tools/mk_trace.py <regime> 8`, two passes per regime (wave / issue counters; LDS "
             "counters), sums over the dispatches of the last four calls (tools/r6_nms_prof.sh, tools/rocpd_calls.py).\n",
             "valu_frac = 4 * SQ_ACTIVE_INST_VALU / (1024 SIMDs * GRBM_GUI_ACTIVE per XCD); wait_frac = SQ_WAIT_ANY / SQ_WAVE_CYCLES; "
             "stall = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES; waves/SIMD = 4 * SQ_WAVE_CYCLES / (1024 * cycles); lds_conflict = SQ_LDS_BANK_CONFLICT / "
             "SQ_LDS_IDX_ACTIVE.  `call` = all kernels of the call together.\n"]
    sq_json = {"_source": "profiles/r6_sq.md (tools/r6_nms_prof.sh: rocprofv3 --kernel-trace --pmc, separate passes per regime)"}
    pmc_path = os.path.join(prof, "r6_pmc.json")
    try:
        pmc = json.load(open(pmc_path))
    except Exception:
        try:
            pmc = json.load(open(os.path.join(prof, "r5_pmc.json")))
            pmc["_inherited_from"] = "profiles/r5_pmc.json (kernels outside the 100k NMS call)"
        except Exception:
            pmc = {}
    for r in REGIMES:
        kt, fe, wr, sa, sb = (load(o, r, p) for p in ("kt", "fetch", "write", "sqa", "sqb"))
        key = r.replace("_raw", "")
        if kt and kt.get("calls"):
            ks.append(f"\n## {r}: {kt['per_call']['dispatches']:.0f} dispatches, {kt['per_call']['kernel_us']:.1f} us of kernels, span {kt['per_call']['span_us']:.1f} us per call\n")
            ks.append("| kernel | dispatches / call | avg us | us / call |\n|---|---:|---:|---:|")
            for name, k in sorted(kt["kernels"].items(), key=lambda t: -t[1]["us_per_call"]):
                ks.append(f"| `{name}` | {k['dispatches_per_call']:.2f} | {k['avg_us']:.2f} | {k['us_per_call']:.2f} |")
            try:
                ks.append("\nlast call, launch order:\n\n```\n" + open(os.path.join(o, f"nms100k_{r}_lastcall.txt")).read().rstrip() + "\n```")
            except Exception:
                pass
        if fe and wr and fe.get("calls") and wr.get("calls"):
            fk = fe["per_call"]["counters"].get("FETCH_SIZE", 0.0)
            wk = wr["per_call"]["counters"].get("WRITE_SIZE", 0.0)
            pmc["nms_100k_call_" + key] = int(round(2 * fk * 1024 + wk * 1024))
            pmc["nms_100k_call_" + key + "_raw_kb"] = {"FETCH_SIZE": round(fk, 1), "WRITE_SIZE": round(wk, 1),
                                                      "calibration_copy_FETCH_SIZE": fe.get("calibration_copy_max", {}).get("FETCH_SIZE"),
                                                      "calibration_copy_WRITE_SIZE": wr.get("calibration_copy_max", {}).get("WRITE_SIZE"),
                                                      "per_kernel_bytes": {n: int(round(2 * fe["kernels"][n]["counters_per_call"].get("FETCH_SIZE", 0.0) * 1024 +
                                                                                        wr["kernels"].get(n, {}).get("counters_per_call", {}).get("WRITE_SIZE", 0.0) * 1024))
                                                                           for n in fe["kernels"]}}
        if sa and sa.get("calls"):
            sq_md.append(f"\n## {r}\n")
            sq_md.append("| kernel | dispatches / call | us / call (profiled) | valu_frac | waves/SIMD | busy | stall | wait_frac | lds_conflict |\n|---|---:|---:|---:|---:|---:|---:|---:|---:|")

            def derive(ca, cb):
                cyc = ca.get("GRBM_GUI_ACTIVE", 0.0)
                wc = ca.get("SQ_WAVE_CYCLES", 0.0)
                f = lambda a, b: (a / b) if b else None
                return {"valu_frac": f(4 * ca.get("SQ_ACTIVE_INST_VALU", 0.0), SIMDS * cyc), "waves_per_simd": f(4 * wc, SIMDS * cyc),
                        "busy_frac": f(ca.get("SQ_ACTIVE_INST_ANY", 0.0), wc), "stall_frac": f(ca.get("SQ_WAIT_INST_ANY", 0.0), wc),
                        "wait_frac": f(ca.get("SQ_WAIT_ANY", 0.0), wc),
                        "lds_conflict_frac": f(cb.get("SQ_LDS_BANK_CONFLICT", 0.0), cb.get("SQ_LDS_IDX_ACTIVE", 0.0)) if cb else None}
            fmt = lambda v: "" if v is None else f"{v:.3f}"
            rows = [(n, k["dispatches_per_call"], k["us_per_call"], derive(k["counters_per_call"], (sb or {}).get("kernels", {}).get(n, {}).get("counters_per_call")))
                    for n, k in sa["kernels"].items()]
            rows.sort(key=lambda t: -t[2])
            rows.append(("call", sa["per_call"]["dispatches"], sa["per_call"]["kernel_us"], derive(sa["per_call"]["counters"], (sb or {}).get("per_call", {}).get("counters"))))
            for n, dpc, us, d in rows:
                sq_md.append(f"| `{n}` | {dpc:.2f} | {us:.1f} | {fmt(d['valu_frac'])} | {fmt(d['waves_per_simd'])} | {fmt(d['busy_frac'])} | {fmt(d['stall_frac'])} | {fmt(d['wait_frac'])} | {fmt(d['lds_conflict_frac'])} |")
                ent = {k: (None if v is None else round(v, 4)) for k, v in d.items()}
                ent["us_profiled_per_call"] = round(us, 2)
                ent["dispatches_per_call"] = dpc
                sq_json[("nms_100k_" + key) if n == "call" else ("nms_100k_" + key + ":" + n)] = ent
    open(os.path.join(prof, "r6_nms100k_kernel_stats.md"), "w").write("\n".join(ks) + "\n")
    open(os.path.join(prof, "r6_sq.md"), "w").write("\n".join(sq_md) + "\n")
    json.dump(sq_json, open(os.path.join(prof, "r6_sq.json"), "w"), indent=1)
    pmc["_nms_100k_call_source"] = "tools/r6_nms_prof.sh: rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE, separate passes per regime; sums over all kernels of a call"
    json.dump(pmc, open(pmc_path, "w"), indent=1)
    print("wrote profiles/r6_nms100k_kernel_stats.md, r6_sq.md, r6_sq.json, r6_pmc.json")


if __name__ == "__main__":
    main()
