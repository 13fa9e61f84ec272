#!/bin/bash
# Round-5 diagnostics in one gpurun call: bash tools/r5_diag.sh <tag>  -> gpurun_out/<tag>/
#   regimes.txt        the four 100k regimes (events + stage events)
#   sq.md              SQ counter passes over tools/prof_sq.py (tools/rocpd_sq.py)
#   hiptrace.md        HIP API timeline summary of the in-loop val NMS bucket (tools/rocpd_hiptrace.py)
TAG=${1:-diag}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
export TMPDIR=/tmp
db() { find $1 -name '*.db' | head -1; }
timeout 300 python tools/prof_regimes.py > $O/regimes.txt 2>&1
REPS=3
PA="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE"
PB="SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"
PC="SQ_IFETCH SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_BUSY_CU_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_BRANCH SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT GRBM_GUI_ACTIVE"
i=0
DBS=""
for P in "$PA" "$PB" "$PC"; do
  i=$((i+1))
  rm -rf /tmp/p_sq$i
  SQ_REPS=$REPS timeout 420 rocprofv3 --kernel-trace --pmc $P -d /tmp/p_sq$i -o sq$i -- python tools/prof_sq.py > $O/sq_pass$i.log 2>&1
  D=$(db /tmp/p_sq$i)
  [ -n "$D" ] && DBS="$DBS $D"
done
python tools/rocpd_sq.py "rocprofv3 --kernel-trace --pmc <8 SQ counters + GRBM_GUI_ACTIVE> -- python tools/prof_sq.py (three passes, SQ_REPS=$REPS)" $REPS $DBS > $O/sq.md 2> $O/sq.err
# fallback: the raw per-dispatch values of this project's kernels (small), in case the summariser needs another look
for D in $DBS; do
  python - "$D" >> $O/sq_raw.csv 2>> $O/sq.err <<'PY'
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
k = {d: (n, s, e) for d, n, s, e in cur.execute("select dispatch_id, name, start, end from kernels")}
for d, cn, v in cur.execute("select dispatch_id, counter_name, counter_value from pmc_events"):
    n, s, e = k.get(d, ("?", 0, 0))
    if "obb::" in n:
        print(f"{s},{e - s},{n.split('(')[0][:60].replace(',', ';')},{cn},{v}")
PY
done
rm -rf /tmp/p_hip
timeout 600 rocprofv3 --hip-trace --kernel-trace --memory-copy-trace -d /tmp/p_hip -o hip -- python tools/trace_valbuckets.py 3 > $O/trace_valbuckets.log 2>&1
python tools/rocpd_hiptrace.py "$(db /tmp/p_hip)" 300 300 > $O/hiptrace.md 2> $O/hiptrace.err
# the same loop without the profiler attached (does the stall need the tracer?)
timeout 300 python tools/trace_valbuckets.py 3 > $O/valbuckets_plain.log 2>&1
cat $O/regimes.txt | grep -E "^clustered|^uniform"; head -c 1500 $O/sq.md; grep -E "^loop|nms bucket" $O/trace_valbuckets.log $O/valbuckets_plain.log | cut -c1-400
