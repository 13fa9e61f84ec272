#!/bin/bash
# Development run on the GPU box: bash tools/run_dev.sh <tag> [pytest args...]  -> gpurun_out/<tag>/
#   PYTEST=0 skips the test run, PROF=1 adds a rocprofv3 kernel trace of tools/prof_regimes.py, BENCH=1 runs bench.py,
#   VARIANTS="v1 sb8": also times yolov5_obb_amd/libobb_hip_<variant>.so (compile-time A/B builds)
TAG=${1:-dev}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
export TMPDIR=/tmp
if [ "${PYTEST:-1}" != "0" ]; then
  timeout 1500 python -m pytest -m gpu -q --durations=10 "$@" > $O/pytest.log 2>&1
  echo "pytest rc=$?" >> $O/pytest.log
fi
timeout 300 python tools/prof_regimes.py > $O/regimes.txt 2>&1
for v in $VARIANTS; do
  OBB_HIP_LIB=$R/yolov5_obb_amd/libobb_hip_$v.so timeout 300 python tools/prof_regimes.py > $O/regimes_$v.txt 2>&1
done
OBB_NMS_PHASE_PROF=1 timeout 300 python tools/prof_regimes.py > $O/phases.txt 2>&1
if [ "${PROF:-0}" = "1" ]; then
  rm -rf /tmp/p_kt
  timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/p_kt -o kt -- python tools/prof_regimes.py > $O/kt.log 2>&1
  python tools/rocpd_summary.py "$(find /tmp/p_kt -name '*.db' | head -1)" "rocprofv3 --kernel-trace --stats -- python tools/prof_regimes.py" > $O/kernel_stats.md 2>&1
fi
if [ "${BENCH:-0}" = "1" ]; then
  timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
fi
[ -f $O/pytest.log ] && tail -14 $O/pytest.log; grep -E "^clustered|^uniform" $O/regimes*.txt; [ -f $O/kernel_stats.md ] && head -14 $O/kernel_stats.md
[ -f $O/bench.json ] && cut -c1-300 $O/bench.json
