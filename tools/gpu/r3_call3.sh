#!/bin/bash
# round 3, GPU call 3: cleaned-up step (fused texture finish + Adam, per-cluster prefix, stats always on), suite, flake record on / off the kink
set +e
O=gpurun_out/r3c3
mkdir -p $O
cd "$GRAFT_REPO_ROOT"
export PYTHONUNBUFFERED=1
echo "== plan timeline" ; timeout 300 python tools/plan_timeline.py --out $O/plan_timeline.txt > $O/plan_timeline.log 2>&1 ; echo rc=$?
grep -E "untimed|host enqueue|sum of" $O/plan_timeline.txt; tail -3 $O/plan_timeline.log
echo "== pytest gpu"
timeout 1500 python -m pytest tests -m gpu -q --durations=8 > $O/pytest_gpu.log 2>&1 ; echo rc=$?
tail -25 $O/pytest_gpu.log
echo "== bench"
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err ; echo rc=$?
python -c "import json; d=json.load(open('$O/bench.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['frac_in_step_deferred'], d['roofline']['frac_isolated'], d['roofline']['us_in_step'], d['roofline']['us_in_step_deferred'], d.get('stage_fps'), d.get('cpu_baseline'))"
tail -3 $O/bench.err
echo "== kbench"
timeout 300 python tools/kbench.py > $O/kbench.txt 2>&1 ; echo rc=$?
cat $O/kbench.txt | tail -30
echo "== flake hunt (on the kink / off the kink)"
timeout 600 python tools/fit_flake_hunt.py 40 10 1.0 > $O/flake_1.0.log 2>&1 ; echo rc=$?
timeout 600 python tools/fit_flake_hunt.py 40 10 1.25 > $O/flake_1.25.log 2>&1 ; echo rc=$?
grep -h "runs whose" $O/flake_1.0.log $O/flake_1.25.log
