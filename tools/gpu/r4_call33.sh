#!/bin/bash
# round 4, GPU call 33: bench.py as the driver runs it (parity leg with the texel-channel distribution), shortened
set +e
O=gpurun_out/r4c33
mkdir -p $O
cd "$GRAFT_REPO_ROOT"
export PYTHONUNBUFFERED=1
timeout 500 python bench.py --steps 160 --no-cpu-baseline --no-stage > $O/bench.json 2> $O/bench.err; echo rc=$?
python -c "import json; d=json.load(open('$O/bench.json')); p=d['parity']; print(round(d['ms_per_step'],4), d['value'], 'tex_extra %.2e' % p['grad_rel']['tex_extra'], 'kink', p['l1_kink_pixels'], p['tex_extra_texel_channels'], 'worst', p['worst_grad'], '%.2e' % p['worst_grad_rel'])"
tail -3 $O/bench.err
