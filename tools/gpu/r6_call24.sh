#!/bin/bash
# round 6, call 24: plan edge events bound to the recording kernel's dispatch (hipExtLaunchKernel stop event) and tail events recorded at join time: A/B + tests
set +e
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp PYTHONUNBUFFERED=1
O=gpurun_out/r6c24; mkdir -p $O
one() {  # $1 = VHAP_DEBUG value, $2 = label
  VHAP_DEBUG=$1 timeout 300 python bench.py --no-cpu-baseline --no-stage --no-parity > $O/bench_$2.json 2> $O/bench_$2.err
  python -c "
import json
d=json.load(open('$O/bench_$2.json')); r=d['roofline']
print('$2', round(d['value']), round(d['ms_per_step'],4), 'frac', round(r['frac'],3), 'shipped', round(r['frac_shipped'],3))"
}
for rep in 1 2; do
  one 6291456 old_$rep          # both off: separate records, tails recorded every replay
  one 0 new_$rep
  one 4194304 bound_only_$rep
  one 2097152 lazy_only_$rep
done
echo "== tests that replay plans"
timeout 1500 python -m pytest tests/test_plan_gpu.py tests/test_native_gpu.py tests/test_tex_carry_gpu.py tests/test_dist_gpu.py -q -x 2>&1 | tail -6
R="$GRAFT_REPO_ROOT"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/$O/prof -o step -- python $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-stage --no-parity > $R/$O/bench_rocprof.json 2> $R/$O/rocprof.err ; echo rc=$?
KT=$(ls $R/$O/prof/*/*kernel_trace.csv $R/$O/prof/*kernel_trace.csv 2>/dev/null | head -1)
[ -n "$KT" ] && python $R/tools/trace_stats.py $KT > $R/$O/trace_stats.txt 2>&1 && python $R/tools/step_profile.py $KT --timeline > $R/$O/step_timeline.txt 2>&1
head -4 $R/$O/trace_stats.txt | cut -c1-110
cut -c1-100 $R/$O/step_timeline.txt | sed -n 1,45p
rm -rf $R/$O/prof
