#!/bin/bash
TAG=${1:-r5j}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
export TMPDIR=/tmp
OBB_NMS_PHASE_PROF=1 timeout 300 python tools/prof_regimes.py > $O/phases.txt 2>&1
grep -E "slab set-up" $O/phases.txt | tail -3 | cut -c1-300
grep -E "^clustered|^uniform" $O/phases.txt
