#!/bin/bash
# round 5, GPU call 1: the multi-GPU readiness tests (bench self-launch, world-1 RCCL sharded step, config 5), the plan tests (+ _set_ints
# under FTZ), then bench.py's parity leg with the oracle's zero-residual fix at three trained states, and the plan timeline
set +e
O=gpurun_out/r5c1
mkdir -p $O
cd "$GRAFT_REPO_ROOT"
export PYTHONUNBUFFERED=1
timeout 1500 python -m pytest tests/test_dist_gpu.py tests/test_plan_gpu.py -m gpu -q -x > $O/pytest.log 2>&1; echo rc=$?; tail -15 $O/pytest.log | cut -c1-300
for st in 100 40 160; do
  timeout 400 python bench.py --steps $st --no-cpu-baseline --no-stage > $O/bench_$st.json 2> $O/bench_$st.err
  python -c "import json; d=json.load(open('$O/bench_$st.json')); p=d['parity']; print($st, round(d['ms_per_step'],4), 'frac', round(d['roofline']['frac'],3), 'tex_extra %.2e' % p['grad_rel']['tex_extra'], 'kink', p['l1_kink_pixels'], p['tex_extra_texel_channels'], 'worst', p['worst_grad'], '%.2e' % p['worst_grad_rel'], 'of max-norm %.2e' % p['grad_max_norm'][p['worst_grad']])" | tee -a $O/parity.txt
done
timeout 300 python tools/plan_timeline.py --out $O/plan_timeline.txt > /dev/null 2>&1
grep -E "untimed|bin_build|frame_prep|flame_skin" $O/plan_timeline.txt | grep -v " s[0-9] " | cut -c1-120
