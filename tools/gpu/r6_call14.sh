#!/bin/bash
# round 6, GPU call 14: the three failures of call 13 with their tracebacks; carried-texture tests (level 2); sharded bench
set +e
O=gpurun_out/r6c14
mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_tex_carry_gpu.py -q -x 2>&1 | tail -5
timeout 900 python -m pytest "tests/test_dist_gpu.py::test_one_rank_rccl_sharded_step_matches_unsharded" -q -x 2>&1 | grep -v "^  \|Warning\|warn" | tail -40
timeout 900 python -m pytest tests/test_native_gpu.py -q 2>&1 | tail -15
for tf in 1 0 1 0; do
VHAP_FORCE_DIST=1 VHAP_SHARD_TEX_FIRST=$tf timeout 300 python bench.py --no-cpu-baseline --no-stage --no-parity > $O/bench_sharded_tf$tf.json 2> $O/bench_sharded_tf$tf.err
python -c "
import json
d=json.load(open('$O/bench_sharded_tf$tf.json')); print('sharded world1 tex_first=$tf', round(d['value']), round(d['ms_per_step'],4))"
done
for d in 1 1; do
  timeout 300 python bench.py --no-cpu-baseline --no-stage --no-parity > $O/bench$d.json 2> $O/bench$d.err
  python -c "
import json
d=json.load(open('$O/bench$d.json')); r=d['roofline']
print(round(d['value']), round(d['ms_per_step'],4), 'frac', round(r['frac'],3), 'shipped', round(r['frac_shipped'],3))"
done
