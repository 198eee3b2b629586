#!/bin/bash
# round 4, GPU call 1: full-batch parity tests (g2), stage-loop deferred-join test, bench line with the parity field
set +e
O=gpurun_out/r4c1
mkdir -p $O
R="$GRAFT_REPO_ROOT"
cd "$R"
export PYTHONUNBUFFERED=1
nproc; free -g | head -2
echo "== full-batch parity + stage loop"
timeout 1500 python -m pytest tests/test_parity_fullbatch_gpu.py tests/test_plan_gpu.py -m gpu -q -x --durations=8 > $O/pytest_new.log 2>&1 ; echo rc=$?
tail -15 $O/pytest_new.log
cp gpurun_out/parity_fullbatch_cfg*.txt gpurun_out/plan_deferred_join_stage_loop.txt $O/ 2>/dev/null
echo "== bench (config 2) with parity"
timeout 600 python bench.py > $O/bench_cfg2.json 2> $O/bench_cfg2.err ; echo rc=$?
tail -3 $O/bench_cfg2.err
python -c "import json; d=json.load(open('$O/bench_cfg2.json')); r=d['roofline']; print(d['value'], d['ms_per_step'], r['frac'], r['frac_in_step_deferred'], r['frac_isolated'], d.get('stage_fps',{}).get('value'), d['cpu_baseline']['value']); print(d.get('parity'))"
