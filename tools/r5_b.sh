#!/bin/bash
# bash tools/r5_b.sh <tag>: binding tests + GC A/B of the val loop + short bench (both bindings)
TAG=${1:-r5b}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_binding_gpu.py tests/test_nmsobb_gpu.py tests/test_valpost_gpu.py tests/test_e2e_gpu.py tests/test_chain_gpu.py -m gpu -x -q --durations=5 > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1
timeout 400 python tools/trace_valbuckets.py 3 0 > $O/valbuckets_gc_on.log 2>&1
timeout 400 python tools/trace_valbuckets.py 3 1 > $O/valbuckets_gc_frozen.log 2>&1
timeout 600 python bench.py --no-extras --no-cpu-baseline > $O/bench_compiled.json 2> $O/bench_compiled.err
OBB_BINDING=ctypes timeout 600 python bench.py --no-extras --no-cpu-baseline > $O/bench_ctypes.json 2> $O/bench_ctypes.err
timeout 300 python tools/time_valtail.py > $O/next_rows.txt 2>&1
tail -5 $O/pytest.log; tail -2 $O/smoke.log; grep -E "^loop|collector" $O/valbuckets_gc_on.log $O/valbuckets_gc_frozen.log | cut -c1-330
for f in $O/bench_compiled.json $O/bench_ctypes.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split('/')[-1], d["value"], d["ms_per_step"], d["ms_per_step_without_stage_events"], d["build"]["binding"], {k:v["ms_per_call"] for k,v in d["nms_100k"]["regimes"].items()})
except Exception as e: print("bench parse failed", e)
PY
done
tail -8 $O/next_rows.txt
