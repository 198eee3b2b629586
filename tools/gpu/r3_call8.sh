#!/bin/bash
# round 3, GPU call 8: deferred join active (texture sort chain in the allowed tails), dynamic offsets fixed
set +e
O=gpurun_out/r3c8
mkdir -p $O
R="$GRAFT_REPO_ROOT"
cd "$R"
export PYTHONUNBUFFERED=1
echo "== new tests"
timeout 900 python -m pytest tests/test_plan_gpu.py tests/test_parity_sizes_gpu.py::test_native_step_with_dynamic_offset_matches_oracle tests/test_native_gpu.py tests/test_ingest.py -q > $O/pytest_new.log 2>&1; tail -8 $O/pytest_new.log; grep -n "^E  " $O/pytest_new.log | head -20
echo "== plan timeline (deferred join) / joined"
timeout 300 python tools/plan_timeline.py --out $O/plan_timeline.txt > $O/plan_timeline.log 2>&1 ; echo rc=$?
grep -E "untimed|host enqueue" $O/plan_timeline.txt
VHAP_DEFER_JOIN=0 timeout 300 python tools/plan_timeline.py --out $O/plan_timeline_joined.txt > $O/plan_timeline_joined.log 2>&1 ; echo rc=$?
grep -E "untimed|host enqueue" $O/plan_timeline_joined.txt
echo "== rocprofv3 kernel trace of bench"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o step -- python $R/bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-stage > $R/$O/bench_rocprof.json 2> $R/$O/rocprof.err ; echo rc=$?
cd "$R"
KT=$(ls $O/prof/*kernel_trace.csv 2>/dev/null | head -1)
[ -n "$KT" ] && python tools/step_profile.py $KT > $O/step_per_kernel.txt 2>&1 && python tools/step_profile.py $KT --timeline > $O/step_timeline.txt 2>&1
cp $O/prof/*kernel_stats.csv $O/step_kernel_stats.csv 2>/dev/null
head -4 $O/step_per_kernel.txt
python -c "import json; d=json.load(open('$O/bench_rocprof.json')); r=d['roofline']; print(d['value'], d['ms_per_step'], r['frac'], r['frac_in_step_deferred'])"
rm -rf $O/prof
