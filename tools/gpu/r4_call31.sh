#!/bin/bash
# round 4, GPU call 31: the bench's parity leg at three trained states (is d(tex_extra) = 7e-3 of call 30 a handful of texels or many?)
set +e
O=gpurun_out/r4c31
mkdir -p $O
cd "$GRAFT_REPO_ROOT"
export PYTHONUNBUFFERED=1
for st in 100 40 160; do
  timeout 400 python bench.py --steps $st --no-cpu-baseline --no-stage > $O/bench_$st.json 2> $O/bench.err
  python -c "import json; d=json.load(open('$O/bench_$st.json')); p=d['parity']; print($st, round(d['ms_per_step'],4), 'tex_extra %.2e' % p['grad_rel']['tex_extra'], 'kink', p['l1_kink_pixels'], p['tex_extra_texel_channels'], 'worst', p['worst_grad'], '%.2e' % p['worst_grad_rel'])" | tee -a $O/parity.txt
done
