#!/bin/bash
# round 6, call 30: the shading backward as 16 x 16 tiles of 8 x 8 wave blocks instead of row order -- tests, A/B (VHAP_DEBUG=16777216 = row order), kbench, trace
set +e
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp PYTHONUNBUFFERED=1
O=gpurun_out/r6c30; mkdir -p $O
timeout 900 python -m pytest tests/test_deferred_gpu.py tests/test_native_gpu.py tests/test_fullsize_gpu.py -q -x 2>&1 | tail -5
one() {
  VHAP_DEBUG=$1 timeout 300 python bench.py --config $3 --no-cpu-baseline --no-stage --no-parity > $O/bench_$2.json 2> $O/bench_$2.err
  python -c "
import json
d=json.load(open('$O/bench_$2.json')); r=d['roofline']
print('$2', round(d['value']), round(d['ms_per_step'],4), 'frac', round(r['frac'],3))"
}
for rep in 1 2; do one 16777216 row_$rep 2; one 0 tiled_$rep 2; done
one 16777216 row_cfg3 3; one 0 tiled_cfg3 3; one 16777216 row_cfg4 4; one 0 tiled_cfg4 4
for f in 16777216 0; do timeout 200 python tools/kbench.py --only deferred_shade_bwd --debug-flags $f 2>&1 | grep -i deferred | sed "s/^/flags=$f /"; done
R="$GRAFT_REPO_ROOT"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/$O/prof -o step -- python $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-stage --no-parity > $R/$O/bench_rocprof.json 2> $R/$O/rocprof.err ; echo rc=$?
KT=$(ls $R/$O/prof/*/*kernel_trace.csv $R/$O/prof/*kernel_trace.csv 2>/dev/null | head -1)
[ -n "$KT" ] && python $R/tools/trace_stats.py $KT > $R/$O/trace_stats.txt 2>&1 && python $R/tools/step_profile.py $KT --timeline > $R/$O/step_timeline.txt 2>&1
head -8 $R/$O/trace_stats.txt | cut -c1-110
cut -c1-100 $R/$O/step_timeline.txt | sed -n 1,45p
rm -rf $R/$O/prof
