#!/bin/bash
# bash tools/r5_g.sh <tag>: LPT A/B on the four 100k regimes (development build of the library), exactness at 100k, val tail, val loop
TAG=${1:-r5g}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
export TMPDIR=/tmp
DEVLIB=$R/yolov5_obb_amd/libobb_hip_dev.so
OBB_HIP_LIB=$DEVLIB OBB_NMS_LPT=0 timeout 300 python tools/prof_regimes.py > $O/regimes_lpt0.txt 2>&1
OBB_HIP_LIB=$DEVLIB OBB_NMS_LPT=1 timeout 300 python tools/prof_regimes.py > $O/regimes_lpt1.txt 2>&1
OBB_HIP_LIB=$DEVLIB OBB_NMS_LPT=1 OBB_NMS_PHASE_PROF=1 timeout 300 python tools/prof_regimes.py > $O/phases_lpt1.txt 2>&1
timeout 300 python tools/prof_regimes.py > $O/regimes.txt 2>&1
timeout 1500 python -m pytest tests/test_nms_gpu.py tests/test_valpost_gpu.py tests/test_nmsobb_gpu.py tests/test_e2e_gpu.py tests/test_chain_gpu.py tests/test_binding_gpu.py -m gpu -q --durations=6 > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
timeout 300 python tools/time_valtail.py > $O/valtail.txt 2>&1
timeout 400 python tools/trace_valbuckets.py 4 0 > $O/vb_compiled.log 2>&1
for f in regimes_lpt0 regimes_lpt1 regimes; do echo "== $f"; grep -E "^clustered|^uniform" $O/$f.txt; done
grep -E "cross phases: mean|nms phases, wg0" $O/phases_lpt1.txt | tail -4 | cut -c1-260
tail -8 $O/pytest.log; tail -1 $O/valtail.txt; grep -E "^loop" $O/vb_compiled.log | cut -c1-300
