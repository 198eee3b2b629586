#!/bin/bash
# round 3, GPU call 4: fixed texture finish (fused vs two-pass), camera backward off the chain, sort behind the photometric sum, stage loop
set +e
O=gpurun_out/r3c4
mkdir -p $O
R="$GRAFT_REPO_ROOT"
cd "$R"
export PYTHONUNBUFFERED=1
echo "== plan timeline (fused texture finish + Adam)" ; timeout 300 python tools/plan_timeline.py --out $O/plan_timeline.txt > $O/plan_timeline.log 2>&1 ; echo rc=$?
grep -E "untimed|host enqueue|sum of|tex_prep_bwd|adam_kernel" $O/plan_timeline.txt; tail -2 $O/plan_timeline.log
echo "== plan timeline (two passes)" ; timeout 300 python tools/plan_timeline.py --two-pass-tex --out $O/plan_timeline_two_pass.txt > $O/plan_timeline2.log 2>&1 ; echo rc=$?
grep -E "untimed|sum of|tex_prep_bwd|adam_kernel" $O/plan_timeline_two_pass.txt
echo "== kbench"
timeout 300 python tools/kbench.py --only tex_finish,adam,disturb,raster_shade,photo_fwd,tex_forward > $O/kbench.txt 2>&1 ; echo rc=$?
tail -12 $O/kbench.txt
echo "== pytest gpu"
timeout 1500 python -m pytest tests -m gpu -q --durations=5 > $O/pytest_gpu.log 2>&1 ; echo rc=$?
tail -12 $O/pytest_gpu.log
echo "== bench"
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err ; echo rc=$?
python -c "import json; d=json.load(open('$O/bench.json')); r=d['roofline']; print(d['value'], d['ms_per_step'], r['frac'], r['frac_in_step_deferred'], r['frac_isolated'], r['us_in_step'], r['us_in_step_deferred'], d.get('stage_fps',{}).get('value'), d.get('cpu_baseline'))"
tail -2 $O/bench.err
echo "== rocprofv3 kernel trace of bench"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o step -- python $R/bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-stage > $R/$O/bench_rocprof.json 2> $R/$O/rocprof.err ; echo rc=$?
cd "$R"
ls $O/prof/* | head
KT=$(ls $O/prof/*/*kernel_trace.csv $O/prof/*kernel_trace.csv 2>/dev/null | head -1)
KS=$(ls $O/prof/*/*kernel_stats.csv $O/prof/*kernel_stats.csv 2>/dev/null | head -1)
echo "trace=$KT stats=$KS"
[ -n "$KT" ] && python tools/step_profile.py $KT > $O/step_per_kernel.txt 2>&1 && python tools/step_profile.py $KT --timeline > $O/step_timeline.txt 2>&1
[ -n "$KS" ] && cp $KS $O/step_kernel_stats.csv
head -30 $O/step_per_kernel.txt
rm -rf $O/prof
echo "== flake hunt (on the ridge / off the ridge)"
timeout 600 python tools/fit_flake_hunt.py 40 10 1.0 > $O/flake_ridge.log 2>&1 ; echo rc=$?
timeout 600 python tools/fit_flake_hunt.py 40 10 1.3,1.15,1.0 > $O/flake_off.log 2>&1 ; echo rc=$?
grep -h "runs whose" $O/flake_ridge.log $O/flake_off.log
