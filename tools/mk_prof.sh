#!/bin/bash
# per-regime kernel traces of the phase-kernel NMS path -> gpurun_out/$1/
out=gpurun_out/${1:-r6mk}; mkdir -p $out
export TMPDIR=/tmp
for r in clustered_k300 clustered_k300_18cls clustered_k3000 uniform; do
  rm -rf /tmp/mkp_$r
  rocprofv3 --kernel-trace -d /tmp/mkp_$r -o t -- env OBB_NMS_MK=1 python tools/mk_trace.py $r 5 > $out/trace_$r.log 2>&1
  db=$(find /tmp/mkp_$r -name '*.db' | head -1)
  python tools/mk_calls.py $db > $out/calls_$r.txt 2>&1
  python tools/rocpd_summary.py $db "$r, 5 calls" > $out/summary_$r.md 2>&1
done
