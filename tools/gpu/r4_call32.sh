#!/bin/bash
# round 4, GPU call 32: the bench's parity leg of the tree BEFORE this session's kernel changes (commit 073c14a, built under _old/) at the same three
# trained states as call 31
set +e
O=$GRAFT_REPO_ROOT/gpurun_out/r4c32
mkdir -p $O
cd "$GRAFT_REPO_ROOT/_old"
export PYTHONUNBUFFERED=1
for st in 100 40 160; do
  timeout 400 python bench.py --steps $st --no-cpu-baseline --no-stage > $O/bench_$st.json 2> $O/bench.err
  python -c "import json; d=json.load(open('$O/bench_$st.json')); p=d['parity']; print($st, round(d['ms_per_step'],4), 'tex_extra %.2e' % p['grad_rel']['tex_extra'], 'kink', p['l1_kink_pixels'], 'worst', p['worst_grad'], '%.2e' % p['worst_grad_rel'])" | tee -a $O/parity.txt
done
