#!/bin/bash
# round 6, GPU call 13: sharded step re-measured (fold back inside the pixel plan), then the whole GPU suite
set +e
O=gpurun_out/r6c13
mkdir -p $O
export PYTHONUNBUFFERED=1
for tf in 1 0 1 0; do
VHAP_FORCE_DIST=1 VHAP_SHARD_TEX_FIRST=$tf timeout 300 python bench.py --no-cpu-baseline --no-stage --no-parity > $O/bench_sharded_tf$tf.json 2> $O/bench_sharded_tf$tf.err
python -c "
import json
d=json.load(open('$O/bench_sharded_tf$tf.json')); print('sharded world1 tex_first=$tf', round(d['value']), round(d['ms_per_step'],4))"
done
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -25 | tee $O/pytest_gpu.log
