#!/bin/bash
# round 4, GPU call 10: buffer-based deferred-join decision (plan tests), whole GPU suite except the full-batch oracle tests, bench line
set +e
O=gpurun_out/r4c10
mkdir -p $O
cd "$GRAFT_REPO_ROOT"
export PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_plan_gpu.py -m gpu -q -x > $O/pytest_plan.log 2>&1; echo rc=$?; tail -5 $O/pytest_plan.log
cp gpurun_out/plan_deferred_join.txt $O/ 2>/dev/null
timeout 1500 python -m pytest tests -m gpu -q --deselect tests/test_parity_fullbatch_gpu.py --durations=5 > $O/pytest_gpu.log 2>&1; echo rc=$?; tail -12 $O/pytest_gpu.log
timeout 600 python bench.py --no-parity > $O/bench_cfg2.json 2> $O/bench_cfg2.err ; echo rc=$?
python -c "import json; d=json.load(open('$O/bench_cfg2.json')); r=d['roofline']; print(d['value'], d['ms_per_step'], r['frac'], r['frac_in_step_deferred'], r['frac_isolated'], d.get('stage_fps',{}).get('value'), d.get('stage_fps',{}).get('steps'), d['cpu_baseline']['value'])"
