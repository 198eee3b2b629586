#!/bin/bash
# round 6, GPU call 8: G-buffer backward with same-triangle lanes merged before the LDS table (debug flag 1024 = without): tests, the
# kernel alone, the step
set +e
O=gpurun_out/r6c8
mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_native_gpu.py tests/test_fused_gpu.py -q -x 2>&1 | tail -6
for d in 0 1024 0 1024; do
  VHAP_DEBUG=$d timeout 300 python tools/kbench.py --only gbuffer_bwd 2>&1 | grep -i "gbuffer" | sed "s/^/dbg=$d /"
done
for d in 0 1024 0 1024; do
  VHAP_DEBUG=$d timeout 300 python bench.py --no-cpu-baseline --no-stage --no-parity > $O/bench$d.json 2> $O/bench$d.err
  python -c "
import json
d=json.load(open('$O/bench$d.json')); r=d['roofline']
print('dbg=$d', round(d['value']), round(d['ms_per_step'],4), 'frac', round(r['frac'],3), 'shipped', round(r['frac_shipped'],3))"
done
R="$GRAFT_REPO_ROOT"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/$O/prof -o step -- python $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-stage --no-parity > $R/$O/bench_rocprof.json 2> $R/$O/rocprof.err ; echo rc=$?
KT=$(ls $R/$O/prof/*kernel_trace.csv 2>/dev/null | head -1)
[ -n "$KT" ] && python $R/tools/trace_stats.py $KT > $R/$O/trace_stats.txt 2>&1 && python $R/tools/step_profile.py $KT --timeline > $R/$O/step_timeline.txt 2>&1
head -14 $R/$O/trace_stats.txt | cut -c1-110
rm -rf $R/$O/prof
