#!/bin/bash
# Round-6 profiles of the single-list rotated NMS at 100k, per regime on the path the library chooses for it:
#   kernel trace, FETCH_SIZE pass, WRITE_SIZE pass, two SQ counter passes (all separate rocprofv3 runs: --pmc never shares a run with
#   anything but --kernel-trace) -> gpurun_out/$1/nms100k_<regime>_<pass>.json (tools/rocpd_calls.py), then tools/r6_collect.py
#   writes profiles/r6_nms100k_kernel_stats.md, profiles/r6_sq.md / .json and the nms_100k_call_* keys of profiles/r6_pmc.json.
TAG=${1:-r6prof}; O=gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
SQA="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE"
SQB="SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"
db() { find $1 -name '*.db' | head -1; }
# regime:path -- the path the un-profiled library chooses for the regime (bench.py's nms_100k times; 1 = the phase kernels of nms_mk.h,
# 0 = the persistent kernel).  Pinned here: the library chooses by the device time of its previous calls, and the profiler's per-dispatch
# overhead (13-21 dispatches against 7) makes it choose differently under rocprofv3.
for spec in clustered_k300_raw:1 clustered_k300_18cls:0 clustered_k3000:1 clustered_k3000_18cls:0 uniform:1; do
  r=${spec%%:*}; mk=${spec##*:}
  for pass in kt fetch write sqa sqb; do
    case $pass in
      kt) PM="";; fetch) PM="--pmc FETCH_SIZE";; write) PM="--pmc WRITE_SIZE";; sqa) PM="--pmc $SQA";; sqb) PM="--pmc $SQB";;
    esac
    D=/tmp/r6p_${r}_$pass; rm -rf $D
    MK_CALIB=1 timeout 300 rocprofv3 --kernel-trace $PM -d $D -o t -- env OBB_NMS_MK=$mk python tools/mk_trace.py $r 8 > $O/nms100k_${r}_$pass.log 2>&1
    python tools/rocpd_calls.py "$(db $D)" 4 > $O/nms100k_${r}_$pass.json 2>> $O/nms100k_${r}_$pass.log
    [ $pass = kt ] && python tools/mk_calls.py "$(db $D)" > $O/nms100k_${r}_lastcall.txt 2>&1
  done
done
python tools/r6_collect.py $O
