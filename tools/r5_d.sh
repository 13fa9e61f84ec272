#!/bin/bash
# bash tools/r5_d.sh <tag>: tests after the sub-pixel-box change + where the 70-88 ms stalls of the ctypes binding come from
TAG=${1:-r5d}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_binding_gpu.py tests/test_nmsobb_gpu.py tests/test_valpost_gpu.py tests/test_e2e_gpu.py tests/test_chain_gpu.py -m gpu -q --durations=5 > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
cg() { echo "--- $1"; cat /sys/fs/cgroup/cpu.max 2>/dev/null; cat /sys/fs/cgroup/cpu.stat 2>/dev/null | grep -E "nr_periods|nr_throttled|throttled_usec"; cat /sys/fs/cgroup/cpu/cpu.cfs_quota_us /sys/fs/cgroup/cpu/cpu.cfs_period_us 2>/dev/null; cat /sys/fs/cgroup/cpu/cpu.stat 2>/dev/null; }
{
  echo "nproc $(nproc)  OMP_NUM_THREADS=${OMP_NUM_THREADS:-unset}"; python -c "import torch; print('torch threads', torch.get_num_threads())"
  cg "before the ctypes loop (default threads)"
  OBB_BINDING=ctypes timeout 400 python tools/trace_valbuckets.py 3 0 > $O/vb_ctypes_default.log 2>&1
  cg "after the ctypes loop (default threads)"
  OMP_NUM_THREADS=1 OBB_BINDING=ctypes timeout 400 python tools/trace_valbuckets.py 3 0 > $O/vb_ctypes_omp1.log 2>&1
  cg "after the ctypes loop with OMP_NUM_THREADS=1"
  timeout 400 python tools/trace_valbuckets.py 3 0 > $O/vb_compiled.log 2>&1
  cg "after the compiled loop"
} > $O/cgroup.txt 2>&1
tail -6 $O/pytest.log; cat $O/cgroup.txt; grep -E "^loop|per batch" $O/vb_*.log | cut -c1-380
