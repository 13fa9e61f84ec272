#!/bin/bash
# per-kernel trace of ONE regime on the phase-kernel path -> gpurun_out/$1/calls_$2.txt   (bash tools/mk_prof1.sh <tag> <regime> [extra env...])
out=gpurun_out/${1:-r6mk}; r=${2:-clustered_k3000_18cls}; shift; shift
mkdir -p $out; export TMPDIR=/tmp
rm -rf /tmp/mkp_$r
rocprofv3 --kernel-trace -d /tmp/mkp_$r -o t -- env OBB_NMS_MK=1 "$@" python tools/mk_trace.py $r 5 > $out/trace_$r.log 2>&1
db=$(find /tmp/mkp_$r -name '*.db' | head -1)
python tools/mk_calls.py $db > $out/calls_$r.txt 2>&1
cat $out/calls_$r.txt
