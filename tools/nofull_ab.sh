#!/bin/bash
O=gpurun_out/${1:-nofull}; mkdir -p $O
for v in 0 1; do for mk in 1; do
  echo "== OBB_NMS_MK_NOFULL=$v" >> $O/t.txt
  OBB_NMS_MK=$mk OBB_NMS_MK_NOFULL=$v timeout 600 python tools/mk_time.py 2>&1 | grep -v amdgpu >> $O/t.txt
done; done
cat $O/t.txt
