#!/bin/bash
# round 3, GPU call 2: whole suite at the shipped defaults (no -x), flake hunt v2, timeline with the 4-kernel disturbance
set +e
O=gpurun_out/r3c2
mkdir -p $O
cd "$GRAFT_REPO_ROOT"
export PYTHONUNBUFFERED=1
echo "== flake hunt"
timeout 900 python tools/fit_flake_hunt.py 60 10 > $O/flake.log 2>&1 ; echo rc=$?
tail -3 $O/flake.log
echo "== plan timeline" ; timeout 300 python tools/plan_timeline.py --out $O/plan_timeline.txt > $O/plan_timeline.log 2>&1 ; echo rc=$?
grep -E "untimed|host enqueue|disturb|sum of" $O/plan_timeline.txt
echo "== pytest gpu (shipped defaults)"
timeout 1500 python -m pytest tests -m gpu -q --durations=15 > $O/pytest_gpu.log 2>&1 ; echo rc=$?
tail -40 $O/pytest_gpu.log
