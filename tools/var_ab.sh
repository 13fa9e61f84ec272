#!/bin/bash
# step time of compile-time variants yolov5_obb_amd/libobb_hip_v<tag>.so against the in-tree build, interleaved twice
R=$(pwd)
for rep in 1 2; do
for v in base $VARIANTS; do
  if [ $v = base ]; then L=$R/yolov5_obb_amd/libobb_hip.so; else L=$R/yolov5_obb_amd/libobb_hip_v$v.so; fi
  echo -n "$v: "; OBB_HIP_LIB=$L python tools/step_time.py 3 2>&1 | grep -v amdgpu | cut -c1-90
done; done
