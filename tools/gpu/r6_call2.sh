#!/bin/bash
# round 6, GPU call 2: carried texture after the contraction fix -- tests, then rocprofv3 kernel traces of the steady-state step with / without
set +e
O=gpurun_out/r6c2
mkdir -p $O
R="$GRAFT_REPO_ROOT"
export PYTHONUNBUFFERED=1
echo "== tests: carried texture"
timeout 600 python -m pytest tests/test_tex_carry_gpu.py -x -q 2>&1 | tail -15
echo "== tests: plan"
timeout 900 python -m pytest tests/test_plan_gpu.py -x -q 2>&1 | tail -8
cd /tmp && export TMPDIR=/tmp
for c in 0 1; do
  VHAP_TEX_CARRY=$c timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof$c -o step -- python $R/bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-stage --no-parity > $R/$O/bench_rocprof$c.json 2> $R/$O/rocprof$c.err ; echo rc=$?
  KT=$(ls $R/$O/prof$c/*kernel_trace.csv 2>/dev/null | head -1)
  [ -n "$KT" ] && python $R/tools/step_profile.py $KT > $R/$O/step_per_kernel_carry$c.txt 2>&1 && python $R/tools/step_profile.py $KT --timeline > $R/$O/step_timeline_carry$c.txt 2>&1
  rm -rf $R/$O/prof$c
done
cd $R
for c in 0 1 0 1; do
  VHAP_TEX_CARRY=$c timeout 300 python bench.py --no-cpu-baseline --no-stage --no-parity > $O/bench_carry$c.json 2> $O/bench_carry$c.err
  python -c "
import json
d=json.load(open('$O/bench_carry$c.json')); r=d['roofline']
print('carry=$c', round(d['value']), round(d['ms_per_step'],4), 'frac', round(r['frac'],3), 'shipped', round(r['frac_shipped'],3))"
done
