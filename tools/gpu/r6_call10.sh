#!/bin/bash
# round 6, GPU call 10: aa_detect with 256-thread workgroups -- tests, bench, timeline
set +e
O=gpurun_out/r6c10
mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_deferred_gpu.py tests/test_fullsize_gpu.py -q -x 2>&1 | tail -6
for d in 1 1 1; do
  timeout 300 python bench.py --no-cpu-baseline --no-stage --no-parity > $O/bench$d.json 2> $O/bench$d.err
  python -c "
import json
d=json.load(open('$O/bench$d.json')); r=d['roofline']
print(round(d['value']), round(d['ms_per_step'],4), 'frac', round(r['frac'],3), 'shipped', round(r['frac_shipped'],3), r['us_in_step_deferred'])"
done
R="$GRAFT_REPO_ROOT"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/$O/prof -o step -- python $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-stage --no-parity > $R/$O/bench_rocprof.json 2> $R/$O/rocprof.err ; echo rc=$?
KT=$(ls $R/$O/prof/*kernel_trace.csv 2>/dev/null | head -1)
[ -n "$KT" ] && python $R/tools/trace_stats.py $KT > $R/$O/trace_stats.txt 2>&1 && python $R/tools/step_profile.py $KT --timeline > $R/$O/step_timeline.txt 2>&1
head -12 $R/$O/trace_stats.txt | cut -c1-110
cut -c1-100 $R/$O/step_timeline.txt | sed -n 1,45p
rm -rf $R/$O/prof
