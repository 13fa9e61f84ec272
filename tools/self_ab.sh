#!/bin/bash
# A/B of the self-sorting segments (OBB_NMS_SELF_SORT) on the headline step and smaller batches -> gpurun_out/$1/
TAG=${1:-self}; O=gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest -m gpu -q -x tests/test_nmsobb_gpu.py > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
for bs in 16 8 4 1; do
  for m in 0 1 2; do
    echo "== BS=$bs OBB_NMS_SELF_SORT=$m" >> $O/step.txt
    BS=$bs OBB_NMS_SELF_SORT=$m timeout 300 python tools/step_time.py 5 2>&1 | grep -v amdgpu.ids >> $O/step.txt
  done
done
tail -5 $O/pytest.log; cat $O/step.txt
