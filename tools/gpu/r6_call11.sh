#!/bin/bash
# round 6, GPU call 11: where the sharded step stands (world-size-1 RCCL group), configs 3 / 4, the 20-step CPU baseline record
set +e
O=gpurun_out/r6c11
mkdir -p $O
export PYTHONUNBUFFERED=1
for tf in 1 0; do
VHAP_FORCE_DIST=1 VHAP_SHARD_TEX_FIRST=$tf timeout 300 python bench.py --no-cpu-baseline --no-stage --no-parity > $O/bench_sharded_tf$tf.json 2> $O/bench_sharded_tf$tf.err
python -c "
import json
d=json.load(open('$O/bench_sharded_tf$tf.json')); print('sharded world1 tex_first=$tf', round(d['value']), round(d['ms_per_step'],4))"
done
for c in 3 4; do
timeout 600 python bench.py --config $c --no-cpu-baseline --no-stage --no-parity > $O/bench_cfg$c.json 2> $O/bench_cfg$c.err
python -c "
import json
d=json.load(open('$O/bench_cfg$c.json')); r=d['roofline']; print('cfg$c', round(d['value']), round(d['ms_per_step'],4), round(r['frac'],3), r.get('frac_shipped'), round(r['frac_isolated'],3))"
done
R="$GRAFT_REPO_ROOT"
cd /tmp && export TMPDIR=/tmp
VHAP_FORCE_DIST=1 VHAP_SHARD_TEX_FIRST=1 timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/$O/prof -o step -- python $R/bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-stage --no-parity > $R/$O/bench_rocprof.json 2> $R/$O/rocprof.err ; echo rc=$?
KT=$(ls $R/$O/prof/*kernel_trace.csv 2>/dev/null | head -1)
[ -n "$KT" ] && python $R/tools/trace_stats.py $KT > $R/$O/trace_stats_sharded.txt 2>&1 && python $R/tools/step_profile.py $KT --timeline > $R/$O/step_timeline_sharded.txt 2>&1
rm -rf $R/$O/prof
cd $R
echo "== bench with the 20-step CPU baseline"
timeout 1500 python bench.py --cpu-baseline-steps 20 --cpu-baseline-warmup 3 > $O/bench_cfg2_cpu20.json 2> $O/bench_cfg2_cpu20.err ; echo rc=$?
python -c "
import json
d=json.load(open('$O/bench_cfg2_cpu20.json')); r=d['roofline']; p=d['parity']; c=d['cpu_baseline']
print(d['value'], d['ms_per_step'], 'frac', r['frac'], 'shipped', r['frac_shipped'], 'iso', r['frac_isolated'], 'stage', (d.get('stage_fps') or {}).get('value'))
print('cpu', c['value'], c['protocol'])
print({k: p.get(k) for k in ('energy_rel','worst_term_rel','worst_grad','worst_grad_rel','l1_kink_pixels')})"
