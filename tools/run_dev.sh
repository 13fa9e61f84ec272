#!/bin/bash
# Development run on the GPU box: bash tools/run_dev.sh <tag> [pytest args...]  -> gpurun_out/<tag>/
TAG=${1:-dev}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x --durations=15 "$@" > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
timeout 300 python tools/prof_regimes.py > $O/regimes.txt 2>&1
OBB_NMS_PHASE_PROF=1 timeout 300 python tools/prof_regimes.py > $O/phases.txt 2>&1
tail -25 $O/pytest.log; grep -E "^clustered|^uniform" $O/regimes.txt
