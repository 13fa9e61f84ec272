#!/bin/bash
# The three SQ counter passes of tools/final_run.sh on their own:  bash tools/sq_run.sh <tag> [round]  -> gpurun_out/<tag>/sq.{md,json} (+ profiles/<round>_sq.*)
TAG=${1:-sq}
RND=${2:-r5}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
export TMPDIR=/tmp
db() { find $1 -name '*.db' | head -1; }
PA="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE"
PB="SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"
PC="SQ_IFETCH SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_BUSY_CU_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_BRANCH SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT GRBM_GUI_ACTIVE"
i=0; DBS=""
for P in "$PA" "$PB" "$PC"; do
  i=$((i+1)); rm -rf /tmp/p_sq$i
  SQ_REPS=3 timeout 420 rocprofv3 --kernel-trace --pmc $P -d /tmp/p_sq$i -o sq$i -- python tools/prof_sq.py > $O/sq_pass$i.log 2>&1
  D=$(db /tmp/p_sq$i); [ -n "$D" ] && DBS="$DBS $D"
done
python tools/rocpd_sq.py "rocprofv3 --kernel-trace --pmc <8 SQ counters + GRBM_GUI_ACTIVE> -- python tools/prof_sq.py (three passes, SQ_REPS=3)" 3 $DBS > $O/sq.md 2> $O/sq.err
python tools/sq_json.py $O/sq.md > $O/sq.json 2>> $O/sq.err && cp $O/sq.json profiles/${RND}_sq.json && cp $O/sq.md profiles/${RND}_sq.md
grep -c "^## " $O/sq.md; cat $O/sq.err | tail -3
