#!/bin/bash
# round 3, GPU call 18: antialias backward off the pixel chain (colour part beside the photometric sum, position part on the side chain)
set +e
O=gpurun_out/r3c18
mkdir -p $O
R="$GRAFT_REPO_ROOT"
cd "$R"
export PYTHONUNBUFFERED=1
echo "== quick parity"
timeout 600 python -m pytest tests/test_native_gpu.py tests/test_energy_gpu.py tests/test_deferred_gpu.py -q -x > $O/pytest_quick.log 2>&1; tail -4 $O/pytest_quick.log; grep -n "^E  " $O/pytest_quick.log | head -10
echo "== plan timeline"
timeout 300 python tools/plan_timeline.py --out $O/plan_timeline.txt > $O/plan_timeline.log 2>&1 ; echo rc=$?
grep -E "untimed|host enqueue" $O/plan_timeline.txt; grep -E "photo_fwd|deferred_shade_bwd|aa_photo_bwd" $O/plan_timeline.txt | cut -c1-100
echo "== pytest gpu (whole suite)"
timeout 1500 python -m pytest tests -m gpu -q --durations=3 > $O/pytest_gpu.log 2>&1 ; echo rc=$?
tail -5 $O/pytest_gpu.log; grep -n "^E  " $O/pytest_gpu.log | head -10
echo "== bench"
timeout 600 python bench.py --no-cpu-baseline --no-stage > $O/bench.json 2> $O/bench.err ; echo rc=$?
python -c "import json; d=json.load(open('$O/bench.json')); r=d['roofline']; print(d['value'], d['ms_per_step'], r['frac'], r['frac_in_step_deferred'], r['frac_isolated'])"
