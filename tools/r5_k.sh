#!/bin/bash
TAG=${1:-r5k}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
export TMPDIR=/tmp
for i in 1 2; do
  timeout 300 python tools/prof_regimes.py > $O/regimes_prod$i.txt 2>&1; echo "== prod $i"; grep -E "^clustered_k300" $O/regimes_prod$i.txt
  OBB_HIP_LIB=$R/yolov5_obb_amd/libobb_hip_ldg.so timeout 300 python tools/prof_regimes.py > $O/regimes_ldg$i.txt 2>&1; echo "== ldg $i"; grep -E "^clustered_k300" $O/regimes_ldg$i.txt
done
OBB_NMS_PHASE_PROF=1 timeout 300 python tools/prof_regimes.py > $O/phases.txt 2>&1
grep "slab set-up (wg0)" $O/phases.txt | tail -2 | cut -c1-300
OBB_HIP_LIB=$R/yolov5_obb_amd/libobb_hip_ldg.so OBB_NMS_PHASE_PROF=1 timeout 300 python tools/prof_regimes.py > $O/phases_ldg.txt 2>&1
grep "slab set-up (wg0)" $O/phases_ldg.txt | tail -2 | cut -c1-300
timeout 600 python -m pytest tests/test_nms_gpu.py -m gpu -q -k "slab or 100k_exact" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
