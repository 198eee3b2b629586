#!/bin/bash
# round 3, GPU call 22: last records of the final code -- bench line of config 2 (with cpu_baseline and stage_fps), kernel trace
set +e
O=gpurun_out/r3c22
mkdir -p $O
R="$GRAFT_REPO_ROOT"
cd "$R"
export PYTHONUNBUFFERED=1
timeout 240 python bench.py > $O/bench_cfg2.json 2> $O/bench_cfg2.err ; echo rc=$?
python -c "import json; d=json.load(open('$O/bench_cfg2.json')); r=d['roofline']; print(d['value'], d['ms_per_step'], r['frac'], r['frac_in_step_deferred'], r['frac_isolated'], d.get('stage_fps',{}).get('value'), d['cpu_baseline']['value'])"
cd /tmp && export TMPDIR=/tmp
timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o step -- python $R/bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-stage > $R/$O/bench_rocprof.json 2> $R/$O/rocprof.err ; echo rc=$?
cd "$R"
KT=$(ls $O/prof/*kernel_trace.csv 2>/dev/null | head -1)
[ -n "$KT" ] && python tools/step_profile.py $KT > $O/step_per_kernel.txt 2>&1 && python tools/step_profile.py $KT --timeline > $O/step_timeline.txt 2>&1
cp $O/prof/*kernel_stats.csv $O/step_kernel_stats.csv 2>/dev/null
head -3 $O/step_per_kernel.txt
