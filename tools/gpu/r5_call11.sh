#!/bin/bash
# round 5, GPU call 11: collectives on the communication stream, four streams on four hardware queues, geometry plan main chain on the launch stream
set +e
O=gpurun_out/r5c11
mkdir -p $O
R="$GRAFT_REPO_ROOT"
cd "$R"
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_dist_gpu.py tests/test_plan_gpu.py -m gpu -q -x > $O/pytest.log 2>&1; echo rc=$?; tail -5 $O/pytest.log | cut -c1-300
VHAP_FORCE_DIST=1 timeout 300 python bench.py --steps 100 --no-cpu-baseline --no-stage --no-parity > $O/bench_forced.json 2> $O/bench_forced.err
timeout 300 python bench.py --steps 100 --no-cpu-baseline --no-parity --no-stage > $O/bench_plain.json 2> $O/bench_plain.err
python -c "
import json
for n in ('forced','plain'):
    d=json.load(open('$O/bench_%s.json' % n)); print(n, round(d['ms_per_step'],4), 'frac', d['roofline']['frac'], 'stage', (d.get('stage_fps') or {}).get('value'))
" | tee -a $O/summary.txt
cd /tmp && export TMPDIR=/tmp
VHAP_FORCE_DIST=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o step -- python $R/bench.py --steps 60 --warmup 20 --no-cpu-baseline --no-stage --no-parity > $R/$O/bench_forced_rocprof.json 2> $R/$O/rocprof.err ; echo rc=$?
cd "$R"
KT=$(ls $O/prof/*kernel_trace.csv 2>/dev/null | head -1)
[ -n "$KT" ] && python tools/step_profile.py $KT --timeline > $O/sharded_step_timeline.txt 2>&1
rm -rf $O/prof
cat $O/sharded_step_timeline.txt | cut -c1-100 | sed -n 1,2p; sed -n 25,60p $O/sharded_step_timeline.txt | cut -c1-100
