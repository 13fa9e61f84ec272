#!/bin/bash
# phase-kernel NMS: parity tests, then times per regime / path, then per-kernel traces -> gpurun_out/$1/
out=gpurun_out/${1:-r6mk}; mkdir -p $out
timeout 900 python -m pytest tests/test_nms_mk_gpu.py -x -q -m gpu > $out/pytest_mk.log 2>&1; tail -3 $out/pytest_mk.log
{
for cfg in "OBB_NMS_MK=0" "OBB_NMS_MK=1 OBB_NMS_MK_XLDS=0" "OBB_NMS_MK=1 OBB_NMS_MK_XLDS=1" "OBB_NMS_MK=2"; do
  env $cfg python tools/mk_time.py 2>&1 | grep -v amdgpu.ids
done
} > $out/times.txt 2>&1
cat $out/times.txt
export TMPDIR=/tmp
bash tools/mk_prof.sh ${1:-r6mk}
