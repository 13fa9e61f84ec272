#!/bin/bash
# serial-phase timers of the phase-kernel NMS path for the four regimes -> gpurun_out/$1/phases.txt
out=gpurun_out/${1:-r6mk}; mkdir -p $out
for r in clustered_k300 clustered_k300_18cls clustered_k3000 uniform; do
  echo "== $r" >> $out/phases.txt
  OBB_NMS_MK=1 OBB_NMS_PHASE_PROF=1 python tools/mk_trace.py $r 4 2>&1 | grep -v amdgpu.ids | tail -7 >> $out/phases.txt
done
