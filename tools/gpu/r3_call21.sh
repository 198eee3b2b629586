#!/bin/bash
# round 3, GPU call 21: the antialias job inside the sum's workgroup budget
set +e
O=gpurun_out/r3c21
mkdir -p $O
cd "$GRAFT_REPO_ROOT"
export PYTHONUNBUFFERED=1
timeout 300 python tools/plan_timeline.py --out $O/plan_timeline.txt > $O/plan_timeline.log 2>&1 ; echo rc=$?
grep -E "untimed|host enqueue" $O/plan_timeline.txt; grep -E "photo_fwd|deferred_shade_bwd|aa_photo_bwd" $O/plan_timeline.txt | tail -3 | cut -c1-100
timeout 600 python -m pytest tests/test_native_gpu.py tests/test_energy_gpu.py tests/test_deferred_gpu.py tests/test_plan_gpu.py -q > $O/pytest_quick.log 2>&1; tail -3 $O/pytest_quick.log; grep -n "^E  " $O/pytest_quick.log | head -10
timeout 600 python bench.py --no-cpu-baseline --no-stage > $O/bench.json 2> $O/bench.err ; echo rc=$?
python -c "import json; d=json.load(open('$O/bench.json')); r=d['roofline']; print(d['value'], d['ms_per_step'], r['frac'], r['frac_in_step_deferred'], r['frac_isolated'])"
