#!/bin/bash
# Full measurement set of a round on the GPU box (run through gpurun from the repo root):
#   bash tools/final_run.sh <tag> [round]   -> gpurun_out/<tag>/{pytest_gpu.log, smoke.log, bench.json, *_kernel_stats.md, pmc_*.md}
# and, on the box, profiles/<round>_pmc.json + <round>_sq_step.{md,json} (default round: r6; the 100k NMS regimes: tools/r6_nms_prof.sh -> <round>_sq.{md,json}) so that the bench line of the same call reads this run's counters.
# rocprofv3 writes rocpd SQLite databases (tens of MB): they stay in /tmp, only the markdown summaries come back.
# SKIP_SQ=1: without the three SQ counter passes (profiles/<round>_sq.* stay as they are); SKIP_HIPTRACE=1: without the traced validation loop.
TAG=${1:-run}
RND=${2:-r6}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 1800 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1
db() { find $1 -name '*.db' | head -1; }
rm -rf /tmp/p_bench /tmp/p_kt /tmp/p_fetch /tmp/p_write
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/p_bench -o bench -- python bench.py --steps 50 --warmup 5 --no-cpu-baseline > $O/bench_prof.log 2>&1
python tools/rocpd_summary.py "$(db /tmp/p_bench)" "rocprofv3 --kernel-trace --stats -- python bench.py --steps 50 --warmup 5 --no-cpu-baseline" > $O/bench_kernel_stats.md 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/p_kt -o kt -- python tools/prof_kernels.py > $O/kt.log 2>&1
python tools/rocpd_summary.py "$(db /tmp/p_kt)" "rocprofv3 --kernel-trace --stats -- python tools/prof_kernels.py" > $O/kernel_stats.md 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/p_fetch -o fetch -- python tools/prof_kernels.py > $O/fetch.log 2>&1
python tools/rocpd_pmc.py "$(db /tmp/p_fetch)" "rocprofv3 --kernel-trace --pmc FETCH_SIZE -- python tools/prof_kernels.py (unit: KB; wide coalesced reads are tallied at half their size on gfx950: the 256 MiB calibration copy reads 262144 KB and reports 131072)" > $O/pmc_fetch.md 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/p_write -o write -- python tools/prof_kernels.py > $O/write.log 2>&1
python tools/rocpd_pmc.py "$(db /tmp/p_write)" "rocprofv3 --kernel-trace --pmc WRITE_SIZE -- python tools/prof_kernels.py (unit: KB; the 256 MiB calibration copy reports 262144)" > $O/pmc_write.md 2>&1
# SQ counters (round 5): three passes of eight SQ counters + GRBM_GUI_ACTIVE over tools/prof_sq.py -> sq.md / sq.json
PA="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE"
PB="SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"
PC="SQ_IFETCH SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_BUSY_CU_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_BRANCH SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT GRBM_GUI_ACTIVE"
i=0; DBS=""
[ -n "$SKIP_SQ" ] || for P in "$PA" "$PB" "$PC"; do
  i=$((i+1)); rm -rf /tmp/p_sq$i
  SQ_REPS=3 timeout 420 rocprofv3 --kernel-trace --pmc $P -d /tmp/p_sq$i -o sq$i -- python tools/prof_sq.py > $O/sq_pass$i.log 2>&1
  D=$(db /tmp/p_sq$i); [ -n "$D" ] && DBS="$DBS $D"
done
if [ -z "$SKIP_SQ" ]; then
python tools/rocpd_sq.py "rocprofv3 --kernel-trace --pmc <8 SQ counters + GRBM_GUI_ACTIVE> -- python tools/prof_sq.py (three passes, SQ_REPS=3)" 3 $DBS > $O/sq.md 2> $O/sq.err
python tools/sq_json.py $O/sq.md > $O/sq.json 2>> $O/sq.err && cp $O/sq.json profiles/${RND}_sq_step.json && cp $O/sq.md profiles/${RND}_sq_step.md
fi
# the counters feed bench.py's `traffic` fields: refresh the json before the bench line is produced
python tools/pmc_json.py gpurun_out/$TAG/pmc_fetch.md gpurun_out/$TAG/pmc_write.md gpurun_out/$TAG/kernel_stats.md > $O/pmc.json 2> $O/pmc_json.err && cp $O/pmc.json profiles/${RND}_pmc.json
# the single-list NMS at 100k, per regime: kernel traces, HBM bytes of the whole call, SQ counters (adds its keys to profiles/<round>_pmc.json)
[ -n "$SKIP_NMS100K" ] || bash tools/r6_nms_prof.sh $TAG > $O/r6_nms_prof.log 2>&1
cp profiles/${RND}_*.md profiles/${RND}_*.json $O/ 2>/dev/null
# the HIP API timeline of the validation loop on the ctypes binding with torch's default thread count (the configuration that showed the
# 70-88 ms stalls of rounds 2-4, before its val tail stopped issuing parallel CPU ops) next to the cgroup's throttle counters, and the same
# loop on the compiled binding: profiles/r5_host_stall.md (the measurement that found the cause is gpurun_out/r5d/cgroup.txt, quoted there)
{ echo "cpu.max $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)"; grep -E "nr_throttled|throttled_usec" /sys/fs/cgroup/cpu.stat 2>/dev/null; } > $O/cgroup_before.txt
if [ -z "$SKIP_HIPTRACE" ]; then
rm -rf /tmp/p_hip
OBB_BINDING=ctypes timeout 600 rocprofv3 --hip-trace --kernel-trace -d /tmp/p_hip -o hip -- python tools/trace_valbuckets.py 2 0 > $O/vb_ctypes_traced.log 2>&1
python tools/rocpd_hiptrace.py "$(db /tmp/p_hip)" 1000 1000 900 > $O/hiptrace_ctypes.md 2> $O/hiptrace.err
grep -E "nr_throttled|throttled_usec" /sys/fs/cgroup/cpu.stat > $O/cgroup_after_ctypes.txt 2>/dev/null
fi
timeout 400 python tools/trace_valbuckets.py 5 0 > $O/vb_compiled.log 2>&1
grep -E "nr_throttled|throttled_usec" /sys/fs/cgroup/cpu.stat > $O/cgroup_after_compiled.txt 2>/dev/null
timeout 300 python tools/time_valtail.py > $O/valtail.txt 2>&1
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
OBB_NMS_PHASE_PROF=1 timeout 300 python tools/prof_regimes.py > $O/nms_phases.txt 2>&1
tail -3 $O/pytest_gpu.log; cat $O/smoke.log | tail -2; cut -c1-400 $O/bench.json
