#!/bin/bash
# round 6, call 31: shapes of the shading backward's workgroups / waves: row order, 16x16/8x8, 64x4/64x1, 32x8/32x2, 16x16/16x4 -- alone and in the step, configs 2 / 3 / 4
set +e
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp PYTHONUNBUFFERED=1
O=gpurun_out/r6c31; mkdir -p $O
for c in 2 3 4; do for f in 16777216 0 33554432 67108864 134217728; do
  timeout 200 python tools/kbench.py --config $c --only deferred_shade_bwd --debug-flags $f 2>&1 | grep -i deferred | sed "s/^/cfg$c flags=$f /"
done; done | tee $O/kbench_shapes.txt
one() {
  VHAP_DEBUG=$1 timeout 300 python bench.py --config $3 --no-cpu-baseline --no-stage --no-parity > $O/bench_$2.json 2> $O/bench_$2.err
  python -c "
import json
d=json.load(open('$O/bench_$2.json')); r=d['roofline']
print('$2', round(d['value']), round(d['ms_per_step'],4))"
}
for c in 2 3 4; do for f in 16777216 0 33554432 67108864 134217728; do one $f cfg${c}_f$f $c; done; done | tee $O/bench_shapes.txt
