#!/bin/bash
# round 3, GPU call 1: does the plan executor work (suite at the shipped defaults), what does it cost vs hipGraphLaunch, first look at
# the in-place disturbance, the flake hunt.
set +e
O=gpurun_out/r3c1
mkdir -p $O
cd "$GRAFT_REPO_ROOT"
export PYTHONUNBUFFERED=1
echo "== plan timeline" ; timeout 300 python tools/plan_timeline.py --out $O/plan_timeline.txt > $O/plan_timeline.log 2>&1 ; echo rc=$?
tail -3 $O/plan_timeline.log
echo "== bench plan vs graph"
for ex in plan graph plan graph; do
  VHAP_EXECUTOR=$ex timeout 300 python bench.py --no-cpu-baseline --steps 200 --warmup 20 > $O/bench_$ex.json 2> $O/bench_$ex.err ; echo "$ex rc=$?"
  python -c "import json,sys; d=json.load(open('$O/bench_$ex.json')); print('$ex', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['frac_in_step_deferred'], d['roofline']['frac_isolated'])"
done
echo "== pytest gpu (shipped defaults)"
timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1 ; echo rc=$?
tail -15 $O/pytest_gpu.log
echo "== flake hunt"
timeout 600 python tools/fit_flake_hunt.py 60 10 > $O/flake.log 2>&1 ; echo rc=$?
tail -5 $O/flake.log
