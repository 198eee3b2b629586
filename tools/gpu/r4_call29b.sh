#!/bin/bash
# round 4, GPU call 29b: the normalisation backward of the vertex normals folded into the vertex kernel: the tests that touch it + bench
set +e
O=gpurun_out/r4c29b
mkdir -p $O
cd "$GRAFT_REPO_ROOT"
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_fused_gpu.py tests/test_native_gpu.py tests/test_plan_gpu.py tests/test_parity_sizes_gpu.py -m gpu -q -x > $O/pytest.log 2>&1; echo rc=$?; tail -3 $O/pytest.log | cut -c1-250
for v in 1 2; do
  timeout 300 python bench.py --no-cpu-baseline --no-stage --no-parity > $O/bench_$v.json 2> $O/bench.err
  python -c "import json; d=json.load(open('$O/bench_$v.json')); r=d['roofline']; print(round(d['ms_per_step'],4), round(d['value']), r.get('us_in_step_deferred'), r.get('frac'))" | tee -a $O/ab.txt
done
