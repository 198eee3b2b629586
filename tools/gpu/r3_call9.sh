#!/bin/bash
# round 3, GPU call 9: plan edge events without the system-scope fence (A/B), dynamic-offset test, full suite, bench + trace
set +e
O=gpurun_out/r3c9
mkdir -p $O
R="$GRAFT_REPO_ROOT"
cd "$R"
export PYTHONUNBUFFERED=1
echo "== plan timeline: device-scope edge events (default) / system-fence events (debug flag 65536)"
timeout 300 python tools/plan_timeline.py --out $O/plan_timeline.txt > $O/plan_timeline.log 2>&1 ; echo rc=$?
grep -E "untimed|host enqueue" $O/plan_timeline.txt
timeout 300 python tools/plan_timeline.py --debug-flags 65536 --out $O/plan_timeline_sysfence.txt > $O/plan_timeline_sysfence.log 2>&1 ; echo rc=$?
grep -E "untimed|host enqueue" $O/plan_timeline_sysfence.txt
echo "== pytest gpu"
timeout 1500 python -m pytest tests -m gpu -q --durations=5 > $O/pytest_gpu.log 2>&1 ; echo rc=$?
tail -12 $O/pytest_gpu.log; grep -n "^E  " $O/pytest_gpu.log | head -20
echo "== bench"
timeout 600 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err ; echo rc=$?
python -c "import json; d=json.load(open('$O/bench.json')); r=d['roofline']; print(d['value'], d['ms_per_step'], r['frac'], r['frac_in_step_deferred'], r['frac_isolated'], r['us_in_step'], r['us_in_step_deferred'], d.get('stage_fps',{}).get('value'))"
tail -2 $O/bench.err
echo "== rocprofv3 kernel trace of bench"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o step -- python $R/bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-stage > $R/$O/bench_rocprof.json 2> $R/$O/rocprof.err ; echo rc=$?
cd "$R"
KT=$(ls $O/prof/*kernel_trace.csv 2>/dev/null | head -1)
[ -n "$KT" ] && python tools/step_profile.py $KT > $O/step_per_kernel.txt 2>&1 && python tools/step_profile.py $KT --timeline > $O/step_timeline.txt 2>&1
cp $O/prof/*kernel_stats.csv $O/step_kernel_stats.csv 2>/dev/null
head -4 $O/step_per_kernel.txt
rm -rf $O/prof
