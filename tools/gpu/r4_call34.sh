#!/bin/bash
# round 4, GPU call 34: bench.py as the driver runs it (supervised child, collector off during captures) + the captured-step tests
set +e
O=gpurun_out/r4c34
mkdir -p $O
cd "$GRAFT_REPO_ROOT"
export PYTHONUNBUFFERED=1
timeout 400 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo rc=$?
python -c "import json; d=json.load(open('$O/bench.json')); p=d['parity']; print(round(d['ms_per_step'],4), d['value'], d.get('stage_fps',{}).get('value'), 'tex_extra %.2e' % p['grad_rel']['tex_extra'], 'kink', p['l1_kink_pixels'], p['tex_extra_texel_channels'], 'worst', p['worst_grad'], '%.2e' % p['worst_grad_rel'])"
tail -3 $O/bench.err | cut -c1-200
timeout 300 python -m pytest tests/test_plan_gpu.py -m gpu -q -x > $O/pytest.log 2>&1; echo rc=$?; tail -2 $O/pytest.log | cut -c1-200
