#!/bin/bash
# round 6, GPU call 3: carried texture -- step test; kernel statistics over all steps of a trace (carry 0 / 1 / 1 with low-priority side streams)
set +e
O=gpurun_out/r6c3
mkdir -p $O
R="$GRAFT_REPO_ROOT"
export PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_tex_carry_gpu.py -x -q 2>&1 | tail -8
cd /tmp && export TMPDIR=/tmp
for v in "0 0" "1 0" "1 524288"; do
  set -- $v
  VHAP_TEX_CARRY=$1 VHAP_DEBUG=$2 timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/$O/prof -o step -- python $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-stage --no-parity > $R/$O/bench_rocprof_$1_$2.json 2> $R/$O/rocprof.err ; echo rc=$?
  KT=$(ls $R/$O/prof/*kernel_trace.csv 2>/dev/null | head -1)
  [ -n "$KT" ] && python $R/tools/trace_stats.py $KT > $R/$O/trace_stats_carry$1_dbg$2.txt 2>&1
  head -30 $R/$O/trace_stats_carry$1_dbg$2.txt | cut -c1-110
  rm -rf $R/$O/prof
done
cd $R
for v in "0 0" "1 0" "1 524288" "0 0" "1 0" "1 524288"; do
  set -- $v
  VHAP_TEX_CARRY=$1 VHAP_DEBUG=$2 timeout 300 python bench.py --no-cpu-baseline --no-stage --no-parity > $O/bench.json 2> $O/bench.err
  python -c "
import json
d=json.load(open('$O/bench.json')); r=d['roofline']
print('carry=$1 dbg=$2', round(d['value']), round(d['ms_per_step'],4), 'frac', round(r['frac'],3), 'shipped', round(r['frac_shipped'],3))"
done
