#!/bin/bash
# bash tools/r5_c.sh <tag>: binding tests, the val loop on the ctypes binding with the collector's passes logged, val tail timing
TAG=${1:-r5c}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_binding_gpu.py tests/test_nmsobb_gpu.py tests/test_valpost_gpu.py tests/test_e2e_gpu.py tests/test_chain_gpu.py -m gpu -q --durations=5 > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
OBB_BINDING=ctypes timeout 400 python tools/trace_valbuckets.py 4 0 > $O/valbuckets_ctypes_gc_on.log 2>&1
OBB_BINDING=ctypes timeout 400 python tools/trace_valbuckets.py 4 1 > $O/valbuckets_ctypes_gc_frozen.log 2>&1
timeout 400 python tools/trace_valbuckets.py 4 0 > $O/valbuckets_compiled_gc_on.log 2>&1
timeout 300 python tools/time_valtail.py > $O/valtail_compiled.txt 2>&1
OBB_BINDING=ctypes timeout 300 python tools/time_valtail.py > $O/valtail_ctypes.txt 2>&1
tail -5 $O/pytest.log; grep -E "^loop|collector|per batch" $O/valbuckets_*.log | cut -c1-420; tail -2 $O/valtail_*.txt
