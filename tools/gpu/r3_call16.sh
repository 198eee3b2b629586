#!/bin/bash
# round 3, GPU call 16: final state -- whole suite, bench lines of configs 2/3/4, kernel trace, plan timeline
set +e
O=gpurun_out/r3c16
mkdir -p $O
R="$GRAFT_REPO_ROOT"
cd "$R"
export PYTHONUNBUFFERED=1
echo "== bench, config 2 (the quoted metric), with the CPU baseline"
timeout 900 python bench.py > $O/bench_cfg2.json 2> $O/bench_cfg2.err ; echo rc=$?
python -c "import json; d=json.load(open('$O/bench_cfg2.json')); r=d['roofline']; print(d['value'], d['ms_per_step'], r['frac'], r['frac_in_step_deferred'], r['frac_isolated'], r.get('traffic'), d.get('stage_fps',{}).get('value'), d['cpu_baseline']['value'])"
echo "== bench, configs 3 and 4"
timeout 600 python bench.py --config 3 --no-cpu-baseline --no-stage > $O/bench_cfg3.json 2> $O/bench_cfg3.err ; echo rc=$?
timeout 600 python bench.py --config 4 --no-cpu-baseline --no-stage > $O/bench_cfg4.json 2> $O/bench_cfg4.err ; echo rc=$?
python -c "
import json
for c in (3, 4):
    d=json.load(open('$O/bench_cfg%d.json' % c)); r=d['roofline']; print(c, d['value'], d['ms_per_step'], r['frac'], r['frac_in_step_deferred'], r['frac_isolated'])"
echo "== plan timeline"
timeout 300 python tools/plan_timeline.py --out $O/plan_timeline.txt > $O/plan_timeline.log 2>&1 ; echo rc=$?
grep -E "untimed|host enqueue" $O/plan_timeline.txt
echo "== rocprofv3 kernel trace of bench"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o step -- python $R/bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-stage > $R/$O/bench_rocprof.json 2> $R/$O/rocprof.err ; echo rc=$?
cd "$R"
KT=$(ls $O/prof/*kernel_trace.csv 2>/dev/null | head -1)
[ -n "$KT" ] && python tools/step_profile.py $KT > $O/step_per_kernel.txt 2>&1 && python tools/step_profile.py $KT --timeline > $O/step_timeline.txt 2>&1
cp $O/prof/*kernel_stats.csv $O/step_kernel_stats.csv 2>/dev/null
head -4 $O/step_per_kernel.txt
echo "== smoke"
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
echo "== pytest gpu (whole suite)"
timeout 1500 python -m pytest tests -m gpu -q --durations=5 > $O/pytest_gpu.log 2>&1 ; echo rc=$?
tail -5 $O/pytest_gpu.log; grep -n "^E  " $O/pytest_gpu.log | head -10
