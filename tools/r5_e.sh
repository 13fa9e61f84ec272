#!/bin/bash
# bash tools/r5_e.sh <tag>: the whole GPU suite + smoke + regimes + val tail timing
TAG=${1:-r5e}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q --durations=12 > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1
timeout 300 python tools/prof_regimes.py > $O/regimes.txt 2>&1
OBB_NMS_PHASE_PROF=1 timeout 300 python tools/prof_regimes.py > $O/phases.txt 2>&1
timeout 300 python tools/time_valtail.py > $O/valtail.txt 2>&1
timeout 400 python tools/trace_valbuckets.py 4 0 > $O/vb_compiled.log 2>&1
tail -15 $O/pytest.log; tail -2 $O/smoke.log; grep -E "^clustered|^uniform" $O/regimes.txt; tail -1 $O/valtail.txt; grep -E "^loop" $O/vb_compiled.log | cut -c1-300
grep -E "cross phases: mean|nms phases, wg0" $O/phases.txt | awk 'NR%5==0' | cut -c1-300 | tail -12
