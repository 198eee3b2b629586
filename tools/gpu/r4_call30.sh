#!/bin/bash
# round 4, GPU call 30: the records of the round -- PMC traffic (RI-fwd pass, whole step), bench lines of configs 2/3/4, kernel trace, plan timeline, suite
set +e
O=gpurun_out/r4c30
mkdir -p $O
R="$GRAFT_REPO_ROOT"
cd /tmp && export TMPDIR=/tmp
export PYTHONUNBUFFERED=1
echo "== PMC: RI-fwd pass"
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/$O/ri_fetch -- python $R/tools/ri_fwd_pmc.py > $R/$O/ri_fetch.log 2>&1 ; echo rc=$?
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/$O/ri_write -- python $R/tools/ri_fwd_pmc.py > $R/$O/ri_write.log 2>&1 ; echo rc=$?
python $R/tools/ri_fwd_pmc.py --report $R/$O/ri_fetch $R/$O/ri_write > $R/$O/r04_ri_fwd_pmc.json 2> $R/$O/ri_report.err ; echo rc=$?
head -c 500 $R/$O/r04_ri_fwd_pmc.json; echo
echo "== PMC: whole step"
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/$O/st_fetch -- python $R/tools/step_pmc.py > $R/$O/st_fetch.log 2>&1 ; echo rc=$?
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/$O/st_write -- python $R/tools/step_pmc.py > $R/$O/st_write.log 2>&1 ; echo rc=$?
python $R/tools/step_pmc.py --report $R/$O/st_fetch $R/$O/st_write > $R/$O/r04_step_pmc.json 2> $R/$O/st_report.err ; echo rc=$?
python -c "import json; d=json.load(open('$R/$O/r04_step_pmc.json')); print(d['step_read_MB'], d['step_write_MB'], d['step_total_MB']); [print(k, v) for k, v in list(d['per_kernel_MB_per_step'].items())[:14]]"
rm -rf $R/$O/ri_fetch $R/$O/ri_write $R/$O/st_fetch $R/$O/st_write
cd "$R"
echo "== bench, config 2 (the quoted metric), with parity + CPU baseline + stage"
timeout 900 python bench.py > $O/bench_cfg2.json 2> $O/bench_cfg2.err ; echo rc=$?
python -c "import json; d=json.load(open('$O/bench_cfg2.json')); r=d['roofline']; print(d['value'], d['ms_per_step'], r['frac'], r['frac_in_step_deferred'], r['frac_isolated'], r.get('traffic'), d.get('stage_fps',{}).get('value'), d['cpu_baseline']['value']); p=d['parity']; print({k: p[k] for k in ('energy_rel','worst_term_rel','worst_grad_rel','worst_grad','min_grad_cos','l1_kink_pixels','seconds')})"
echo "== bench, configs 3 and 4"
timeout 600 python bench.py --config 3 --no-cpu-baseline --no-stage --no-parity > $O/bench_cfg3.json 2> $O/bench_cfg3.err ; echo rc=$?
timeout 600 python bench.py --config 4 --no-cpu-baseline --no-stage --no-parity > $O/bench_cfg4.json 2> $O/bench_cfg4.err ; echo rc=$?
python -c "
import json
for c in (3, 4):
    d=json.load(open('$O/bench_cfg%d.json' % c)); r=d['roofline']; print(c, d['value'], d['ms_per_step'], r['frac'], r['frac_in_step_deferred'], r['frac_isolated'])"
echo "== plan timeline"
timeout 300 python tools/plan_timeline.py --out $O/plan_timeline.txt > $O/plan_timeline.log 2>&1 ; echo rc=$?
grep -E "untimed|host enqueue" $O/plan_timeline.txt
echo "== rocprofv3 kernel trace of bench"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o step -- python $R/bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-stage --no-parity > $R/$O/bench_rocprof.json 2> $R/$O/rocprof.err ; echo rc=$?
cd "$R"
KT=$(ls $O/prof/*kernel_trace.csv 2>/dev/null | head -1)
[ -n "$KT" ] && python tools/step_profile.py $KT > $O/step_per_kernel.txt 2>&1 && python tools/step_profile.py $KT --timeline > $O/step_timeline.txt 2>&1
cp $O/prof/*kernel_stats.csv $O/step_kernel_stats.csv 2>/dev/null
head -4 $O/step_per_kernel.txt
rm -rf $O/prof
echo "== kbench"
timeout 300 python tools/kbench.py > $O/kbench.txt 2>&1 ; echo rc=$?
tail -5 $O/kbench.txt
echo "== smoke"
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
echo "== pytest gpu (whole suite)"
timeout 2400 python -m pytest tests -m gpu -q --durations=8 > $O/pytest_gpu.log 2>&1 ; echo rc=$?
tail -14 $O/pytest_gpu.log; grep -n "^E  " $O/pytest_gpu.log | head -10
cp gpurun_out/parity_fullbatch_cfg*.txt gpurun_out/parity_native_injected_cfg*.txt gpurun_out/fit_parity_10_steps_512_T2048.txt $O/ 2>/dev/null
