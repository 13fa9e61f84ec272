#!/bin/bash
for cfg in "OBB_NMS_MK_CHUNK=2048" "OBB_NMS_MK_CHUNK=4096"; do
  echo "== $cfg"
  env OBB_NMS_MK=1 $cfg python tools/prof_regimes.py 2>&1 | grep -v amdgpu
done
