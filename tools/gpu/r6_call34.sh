#!/bin/bash
# round 6, call 34: the shading backward over the list of covered pixels the disturbance's counting sort leaves (VHAP_SHADE_LIST=0: the pass over the frame) -- tests, A/B, trace
set +e
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp PYTHONUNBUFFERED=1
O=gpurun_out/r6c34; mkdir -p $O
timeout 1200 python -m pytest tests/test_deferred_gpu.py tests/test_native_gpu.py tests/test_fullsize_gpu.py tests/test_tex_carry_gpu.py tests/test_plan_gpu.py tests/test_energy_gpu.py -q -x 2>&1 | tail -5
one() {
  VHAP_SHADE_LIST=$1 timeout 300 python bench.py --config $3 --no-cpu-baseline --no-stage --no-parity > $O/bench_$2.json 2> $O/bench_$2.err
  python -c "
import json
d=json.load(open('$O/bench_$2.json')); r=d['roofline']
print('$2', round(d['value']), round(d['ms_per_step'],4))"
}
for rep in 1 2; do for c in 2 3 4; do one 0 frame_cfg${c}_$rep $c; one 1 list_cfg${c}_$rep $c; done; done | tee $O/bench_ab.txt
R="$GRAFT_REPO_ROOT"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/$O/prof -o step -- python $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-stage --no-parity > $R/$O/bench_rocprof.json 2> $R/$O/rocprof.err ; echo rc=$?
KT=$(ls $R/$O/prof/*/*kernel_trace.csv $R/$O/prof/*kernel_trace.csv 2>/dev/null | head -1)
[ -n "$KT" ] && python $R/tools/trace_stats.py $KT > $R/$O/trace_stats.txt 2>&1 && python $R/tools/step_profile.py $KT --timeline > $R/$O/step_timeline.txt 2>&1
head -12 $R/$O/trace_stats.txt | cut -c1-110
cut -c1-100 $R/$O/step_timeline.txt | sed -n 1,45p
rm -rf $R/$O/prof
