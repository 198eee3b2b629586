#!/bin/bash
# round 5, GPU call 12: the texture's way home on the communication stream end to end (reduce-scatter -> row finish + Adam -> all-gather), geometry first / texture first
set +e
O=gpurun_out/r5c12
mkdir -p $O
R="$GRAFT_REPO_ROOT"
cd "$R"
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_dist_gpu.py -m gpu -q -x > $O/pytest.log 2>&1; echo rc=$?; tail -5 $O/pytest.log | cut -c1-300
for tf in 0 1; do
  VHAP_SHARD_TEX_FIRST=$tf VHAP_FORCE_DIST=1 timeout 300 python bench.py --steps 100 --no-cpu-baseline --no-stage --no-parity > $O/bench_forced_tf$tf.json 2> $O/bench_forced_tf$tf.err
  python -c "
import json
d=json.load(open('$O/bench_forced_tf$tf.json')); print('texture first $tf:', round(d['ms_per_step'],4), d['config']['tex_first'], d['config']['tex_sharded'])
" | tee -a $O/summary.txt
done
cd /tmp && export TMPDIR=/tmp
for tf in 0 1; do
VHAP_SHARD_TEX_FIRST=$tf VHAP_FORCE_DIST=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof$tf -o step -- python $R/bench.py --steps 60 --warmup 20 --no-cpu-baseline --no-stage --no-parity > $R/$O/bench_forced_rocprof$tf.json 2> $R/$O/rocprof$tf.err ; echo rc=$?
KT=$(ls $R/$O/prof$tf/*kernel_trace.csv 2>/dev/null | head -1)
[ -n "$KT" ] && python $R/tools/step_profile.py $KT --timeline > $R/$O/sharded_step_timeline_tf$tf.txt 2>&1
rm -rf $R/$O/prof$tf
done
cd "$R"
for tf in 0 1; do sed -n 1p $O/sharded_step_timeline_tf$tf.txt; sed -n 28,60p $O/sharded_step_timeline_tf$tf.txt | cut -c1-100; done
