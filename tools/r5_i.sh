#!/bin/bash
TAG=${1:-r5i}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
export TMPDIR=/tmp
DEVLIB=$R/yolov5_obb_amd/libobb_hip_dev.so
for v in 0 1 3 0 3; do
  OBB_HIP_LIB=$DEVLIB OBB_NMS_LPT=$v timeout 300 python tools/prof_regimes.py > $O/regimes_lpt$v.txt 2>&1
  echo "== lpt $v"; grep -E "^clustered|^uniform" $O/regimes_lpt$v.txt
done
OBB_HIP_LIB=$DEVLIB OBB_NMS_LPT=3 OBB_NMS_PHASE_PROF=1 timeout 300 python tools/prof_regimes.py > $O/phases_lpt3.txt 2>&1
grep -E "cross phases: mean|nms phases, wg0" $O/phases_lpt3.txt | tail -2 | cut -c1-260
timeout 1500 python -m pytest tests/test_nms_gpu.py -m gpu -q --durations=4 > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
tail -5 $O/pytest.log
