#!/bin/bash
# Full measurement set of a round on the GPU box (run through gpurun from the repo root):
#   bash tools/final_run.sh <tag> [round]   -> gpurun_out/<tag>/{pytest_gpu.log, smoke.log, bench.json, *_kernel_stats.md, pmc_*.md}
# and, on the box, profiles/<round>_pmc.json (default round: r4) so that the bench line of the same call reads this run's counters.
# rocprofv3 writes rocpd SQLite databases (tens of MB): they stay in /tmp, only the markdown summaries come back.
TAG=${1:-run}
RND=${2:-r4}
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
# the counters feed bench.py's `traffic` fields: refresh the json before the bench line is produced
python tools/pmc_json.py gpurun_out/$TAG/pmc_fetch.md gpurun_out/$TAG/pmc_write.md gpurun_out/$TAG/kernel_stats.md > $O/pmc.json 2> $O/pmc_json.err && cp $O/pmc.json profiles/${RND}_pmc.json
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
OBB_NMS_PHASE_PROF=1 timeout 300 python tools/prof_regimes.py > $O/nms_phases.txt 2>&1
tail -3 $O/pytest_gpu.log; cat $O/smoke.log | tail -2; cut -c1-400 $O/bench.json
