#!/bin/bash
# round 6, GPU call 4: carried texture with the finish pass held to 80 VGPRs -- test, trace statistics, bench A/B
set +e
O=gpurun_out/r6c4
mkdir -p $O
R="$GRAFT_REPO_ROOT"
export PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_tex_carry_gpu.py -x -q 2>&1 | tail -8
cd /tmp && export TMPDIR=/tmp
VHAP_TEX_CARRY=1 timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/$O/prof -o step -- python $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-stage --no-parity > $R/$O/bench_rocprof.json 2> $R/$O/rocprof.err ; echo rc=$?
KT=$(ls $R/$O/prof/*kernel_trace.csv 2>/dev/null | head -1)
[ -n "$KT" ] && python $R/tools/trace_stats.py $KT > $R/$O/trace_stats_carry1.txt 2>&1 && python $R/tools/step_profile.py $KT --timeline > $R/$O/step_timeline_carry1.txt 2>&1
head -44 $R/$O/trace_stats_carry1.txt | cut -c1-110
rm -rf $R/$O/prof
cd $R
for c in 0 1 0 1; do
  VHAP_TEX_CARRY=$c timeout 300 python bench.py --no-cpu-baseline --no-stage --no-parity > $O/bench$c.json 2> $O/bench$c.err
  python -c "
import json
d=json.load(open('$O/bench$c.json')); r=d['roofline']
print('carry=$c', round(d['value']), round(d['ms_per_step'],4), 'frac', round(r['frac'],3), 'shipped', round(r['frac_shipped'],3))"
done
