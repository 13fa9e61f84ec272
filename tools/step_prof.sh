#!/bin/bash
# kernel durations and launch gaps of the headline step: rocprofv3 --kernel-trace over tools/step_time.py (four rotated tensors)
O=${1:-gpurun_out/stepprof}; mkdir -p $O; export TMPDIR=/tmp; rm -rf /tmp/p_step
timeout 600 rocprofv3 --kernel-trace -d /tmp/p_step -o step -- python tools/step_time.py 2 > $O/step_prof.log 2>&1
python tools/rocpd_timeline.py "$(find /tmp/p_step -name '*.db' | head -1)" 100 | tee $O/step_timeline.md
