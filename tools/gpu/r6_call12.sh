#!/bin/bash
# round 6, GPU call 12: the sharded step with the energy assembly + antialias colour job inside the photometric sum, fold outside the pixel
# plan (tex_first) -- tests, world-size-1 bench both orderings, timelines; kbench cold vs warm
set +e
O=gpurun_out/r6c12
mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 1500 python -m pytest tests/test_dist_gpu.py -q -x 2>&1 | tail -8
R="$GRAFT_REPO_ROOT"
for tf in 1 0 1 0; do
VHAP_FORCE_DIST=1 VHAP_SHARD_TEX_FIRST=$tf timeout 300 python bench.py --no-cpu-baseline --no-stage --no-parity > $O/bench_sharded_tf$tf.json 2> $O/bench_sharded_tf$tf.err
python -c "
import json
d=json.load(open('$O/bench_sharded_tf$tf.json')); print('sharded world1 tex_first=$tf', round(d['value']), round(d['ms_per_step'],4))"
done
cd /tmp && export TMPDIR=/tmp
for tf in 1 0; do
VHAP_FORCE_DIST=1 VHAP_SHARD_TEX_FIRST=$tf timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/$O/prof -o step -- python $R/bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-stage --no-parity > $R/$O/bench_rocprof.json 2> $R/$O/rocprof.err ; echo rc=$?
KT=$(ls $R/$O/prof/*kernel_trace.csv 2>/dev/null | head -1)
[ -n "$KT" ] && python $R/tools/step_profile.py $KT --timeline > $R/$O/step_timeline_sharded_tf$tf.txt 2>&1
rm -rf $R/$O/prof
done
cd $R
echo "== kbench warm / cold"
timeout 600 python tools/kbench.py > $O/kbench_warm.txt 2>&1
timeout 600 python tools/kbench.py --cold --reps 9 > $O/kbench_cold.txt 2>&1
paste -d'|' <(cut -c1-12 $O/kbench_warm.txt) <(cut -c1-110 $O/kbench_cold.txt) | tail -28
