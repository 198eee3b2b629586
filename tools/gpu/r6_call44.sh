#!/bin/bash
# round 6, call 44: the list-driven shading backward held to 4 / 5 waves per SIMD (128 / 96 VGPRs; the compiler's choice is 142 = 3 waves)
set +e
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp PYTHONUNBUFFERED=1
O=gpurun_out/r6c44; mkdir -p $O
one() {
  VHAP_DEBUG=$1 timeout 300 python bench.py --config $3 --no-cpu-baseline --no-stage --no-parity > $O/bench_$2.json 2> $O/bench_$2.err
  python -c "
import json
d=json.load(open('$O/bench_$2.json'))
print('$2', round(d['value']), round(d['ms_per_step'],4))"
}
for rep in 1 2; do one 0 w3_cfg2_$rep 2; one 1073741824 w4_cfg2_$rep 2; one 536870912 w5_cfg2_$rep 2; done | tee $O/ab.txt
for c in 3 4; do one 0 w3_cfg$c $c; one 1073741824 w4_cfg$c $c; one 536870912 w5_cfg$c $c; done | tee -a $O/ab.txt
