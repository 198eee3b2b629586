#!/bin/bash
# round 5, GPU call 13: only the forward plan's texture chain waits for the all-gather (vhap_plan_side_stream_wait)
set +e
O=gpurun_out/r5c13
mkdir -p $O
R="$GRAFT_REPO_ROOT"
cd "$R"
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_dist_gpu.py -m gpu -q -x > $O/pytest.log 2>&1; echo rc=$?; tail -5 $O/pytest.log | cut -c1-300
for pw in 0 1; do for tf in 0 1; do
  VHAP_SHARD_PRECISE_WAIT=$pw VHAP_SHARD_TEX_FIRST=$tf VHAP_FORCE_DIST=1 timeout 300 python bench.py --steps 100 --no-cpu-baseline --no-stage --no-parity > $O/bench_pw${pw}_tf$tf.json 2> $O/bench_pw${pw}_tf$tf.err
  python -c "
import json
d=json.load(open('$O/bench_pw${pw}_tf$tf.json')); print('precise wait $pw, texture first $tf:', round(d['ms_per_step'],4))
" | tee -a $O/summary.txt
done; done
timeout 300 python bench.py --steps 100 --no-cpu-baseline --no-stage --no-parity > $O/bench_plain.json 2> $O/bench_plain.err; python -c "
import json
d=json.load(open('$O/bench_plain.json')); print('one-plan step:', round(d['ms_per_step'],4))" | tee -a $O/summary.txt
