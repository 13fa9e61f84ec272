#!/bin/bash
TAG=${1:-r5f}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_binding_gpu.py tests/test_nmsobb_gpu.py tests/test_valpost_gpu.py tests/test_e2e_gpu.py tests/test_chain_gpu.py -m gpu -q --durations=5 > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
timeout 300 python tools/time_valtail.py > $O/valtail.txt 2>&1
timeout 400 python tools/trace_valbuckets.py 4 0 > $O/vb_compiled.log 2>&1
tail -6 $O/pytest.log; tail -1 $O/valtail.txt; grep -E "^loop|per batch" $O/vb_compiled.log | cut -c1-330
