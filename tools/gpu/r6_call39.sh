#!/bin/bash
# round 6, GPU call 39 (the final tree: the shading backward walks the list of covered pixels): the records of the round -- the whole GPU suite, bench lines of configs 2 / 3 / 4 (+ the sharded step on a world-size-1
# RCCL group), PMC traffic (RI-fwd pass, whole step), kernel trace + statistics, plan timeline, kernel micro-benchmarks
set +e
O=gpurun_out/r6c39
mkdir -p $O
R="$GRAFT_REPO_ROOT"
export PYTHONUNBUFFERED=1
echo "== pytest -m gpu"
timeout 2400 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; tail -12 $O/pytest_gpu.log
cp gpurun_out/parity_fullbatch_cfg2.txt gpurun_out/parity_fullbatch_cfg3.txt gpurun_out/parity_fullbatch_cfg4.txt $O/ 2>/dev/null
echo "== bench, config 2 (the quoted metric), with parity + CPU baseline + stage"
timeout 900 python bench.py > $O/bench_cfg2.json 2> $O/bench_cfg2.err ; echo rc=$?
python -c "
import json
d=json.load(open('$O/bench_cfg2.json')); r=d['roofline']; p=d['parity']
print(d['value'], d['ms_per_step'], 'frac', r['frac'], 'shipped', r['frac_shipped'], 'iso', r['frac_isolated'], 'stage', (d.get('stage_fps') or {}).get('value'), 'cpu', d['cpu_baseline']['value'], d.get('supervisor'))
print(r['us_in_step'], r['us_in_step_deferred'])
print({k: p.get(k) for k in ('energy_rel','worst_term_rel','worst_grad','worst_grad_rel','l1_kink_pixels')})"
echo "== bench, configs 3 and 4; the sharded step on a world-size-1 RCCL group"
timeout 600 python bench.py --config 3 --no-cpu-baseline --no-stage --no-parity > $O/bench_cfg3.json 2> $O/bench_cfg3.err ; echo rc=$?
timeout 600 python bench.py --config 4 --no-cpu-baseline --no-stage --no-parity > $O/bench_cfg4.json 2> $O/bench_cfg4.err ; echo rc=$?
VHAP_FORCE_DIST=1 VHAP_SHARD_TEX_FIRST=1 timeout 300 python bench.py --no-cpu-baseline --no-stage --no-parity > $O/bench_cfg2_sharded_world1.json 2> $O/bench_cfg2_sharded_world1.err ; echo rc=$?
python -c "
import json
for c in ('cfg3', 'cfg4', 'cfg2_sharded_world1'):
    d=json.load(open('$O/bench_%s.json' % c)); r=d['roofline']; print(c, d['value'], d['ms_per_step'], r['frac'], r.get('frac_shipped'), r['frac_isolated'])"
echo "== plan timeline"
timeout 300 python tools/plan_timeline.py --out $O/plan_timeline.txt > $O/plan_timeline.log 2>&1 ; echo rc=$?
grep -E "untimed|host enqueue" $O/plan_timeline.txt
echo "== rocprofv3 kernel trace of bench"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o step -- python $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-stage --no-parity > $R/$O/bench_rocprof.json 2> $R/$O/rocprof.err ; echo rc=$?
cd "$R"
KT=$(ls $O/prof/*kernel_trace.csv 2>/dev/null | head -1)
[ -n "$KT" ] && python tools/step_profile.py $KT > $O/step_per_kernel.txt 2>&1 && python tools/step_profile.py $KT --timeline > $O/step_timeline.txt 2>&1 && python tools/trace_stats.py $KT > $O/trace_stats.txt 2>&1
cp $O/prof/*kernel_stats.csv $O/step_kernel_stats.csv 2>/dev/null
head -44 $O/trace_stats.txt | cut -c1-110
rm -rf $O/prof
cd /tmp
VHAP_FORCE_DIST=1 VHAP_SHARD_TEX_FIRST=1 timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/$O/prof -o step -- python $R/bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-stage --no-parity > $R/$O/bench_rocprof_sharded.json 2> $R/$O/rocprof_sharded.err ; echo rc=$?
cd "$R"
KT=$(ls $O/prof/*kernel_trace.csv 2>/dev/null | head -1)
[ -n "$KT" ] && python tools/step_profile.py $KT --timeline > $O/step_timeline_sharded.txt 2>&1
rm -rf $O/prof
echo "== PMC: RI-fwd pass"
cd /tmp
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/$O/ri_fetch -- python $R/tools/ri_fwd_pmc.py > $R/$O/ri_fetch.log 2>&1 ; echo rc=$?
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/$O/ri_write -- python $R/tools/ri_fwd_pmc.py > $R/$O/ri_write.log 2>&1 ; echo rc=$?
python $R/tools/ri_fwd_pmc.py --report $R/$O/ri_fetch $R/$O/ri_write > $R/$O/r06_ri_fwd_pmc.json 2> $R/$O/ri_report.err ; echo rc=$?
head -c 300 $R/$O/r06_ri_fwd_pmc.json; echo
echo "== PMC: whole step"
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/$O/st_fetch -- python $R/tools/step_pmc.py > $R/$O/st_fetch.log 2>&1 ; echo rc=$?
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/$O/st_write -- python $R/tools/step_pmc.py > $R/$O/st_write.log 2>&1 ; echo rc=$?
python $R/tools/step_pmc.py --report $R/$O/st_fetch $R/$O/st_write > $R/$O/r06_step_pmc.json 2> $R/$O/st_report.err ; echo rc=$?
python -c "import json; d=json.load(open('$R/$O/r06_step_pmc.json')); print(d['step_read_MB'], d['step_write_MB'], d['step_total_MB'])"
rm -rf $R/$O/ri_fetch $R/$O/ri_write $R/$O/st_fetch $R/$O/st_write
cd "$R"
echo "== kbench"
timeout 300 python tools/kbench.py > $O/kbench.txt 2>&1 ; echo rc=$?
for f in 0 1024 256; do VHAP_X=1 timeout 200 python tools/kbench.py --only gbuffer_bwd --debug-flags $f 2>&1 | grep -i gbuffer | sed "s/^/flags=$f /"; done
tail -5 $O/kbench.txt
du -sh gpurun_out
