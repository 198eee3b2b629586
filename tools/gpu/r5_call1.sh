#!/bin/bash
# round 5, GPU call 1 (prepared at the end of round 4, whose GPU budget was spent before these could run):
#   - the whole GPU suite on the final tree of round 4 (two kernel edits were ISA-checked only: texgrad_tile's stand-in address, the pinned vertex-normal gather)
#   - bench.py with the oracle's zero-residual fix: does d(tex_extra) of the parity leg now stay at 1e-5 at trained states?  (three states)
#   - the plan timeline (bin_build with the pinned vertex-normal gather: 24 us before)
set +e
O=gpurun_out/r5c1
mkdir -p $O
cd "$GRAFT_REPO_ROOT"
export PYTHONUNBUFFERED=1
timeout 1800 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo rc=$?; tail -4 $O/pytest.log | cut -c1-250
for st in 100 40 160; do
  timeout 400 python bench.py --steps $st --no-cpu-baseline --no-stage > $O/bench_$st.json 2> $O/bench.err
  python -c "import json; d=json.load(open('$O/bench_$st.json')); p=d['parity']; print($st, round(d['ms_per_step'],4), 'tex_extra %.2e' % p['grad_rel']['tex_extra'], 'kink', p['l1_kink_pixels'], p['tex_extra_texel_channels'], 'worst', p['worst_grad'], '%.2e' % p['worst_grad_rel'], 'of max-norm %.2e' % p['grad_max_norm'][p['worst_grad']])" | tee -a $O/parity.txt
done
timeout 300 python tools/plan_timeline.py --out $O/plan_timeline.txt > /dev/null 2>&1
grep -E "untimed|bin_build|frame_prep|flame_skin" $O/plan_timeline.txt | grep -v " s[0-9] " | cut -c1-120
