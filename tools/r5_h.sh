#!/bin/bash
TAG=${1:-r5h}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 300 python tools/prof_regimes.py > $O/regimes.txt 2>&1
OBB_NMS_PHASE_PROF=1 timeout 300 python tools/prof_regimes.py > $O/phases.txt 2>&1
timeout 1500 python -m pytest tests/test_nms_gpu.py tests/test_nmsobb_gpu.py -m gpu -q --durations=4 > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
timeout 400 python tools/trace_valbuckets.py 4 0 > $O/vb_compiled.log 2>&1
grep -E "^clustered|^uniform" $O/regimes.txt
grep -E "cross phases: mean|nms phases, wg0" $O/phases.txt | tail -2 | cut -c1-260
tail -5 $O/pytest.log; grep -E "^loop" $O/vb_compiled.log | cut -c1-300
