#!/bin/bash
# round 6, GPU call 1: the carried texture (vhap_tex_finish_carry) -- kernel-level and step-level tests, then A/B of the step with and
# without it (bench lines + plan timelines)
set +e
O=gpurun_out/r6c1
mkdir -p $O
export PYTHONUNBUFFERED=1
echo "== tests: carried texture"
timeout 600 python -m pytest tests/test_tex_carry_gpu.py -x -q 2>&1 | tail -15
echo "== tests: plan / native / fit parity"
timeout 900 python -m pytest tests/test_plan_gpu.py tests/test_native_gpu.py tests/test_fit_parity_gpu.py -x -q 2>&1 | tail -15
echo "== bench A/B"
for c in 0 1 0 1; do
  VHAP_TEX_CARRY=$c timeout 300 python bench.py --no-cpu-baseline --no-stage --no-parity > $O/bench_carry$c.json 2> $O/bench_carry$c.err ; echo rc=$?
  python -c "
import json
d=json.load(open('$O/bench_carry$c.json')); r=d['roofline']
print('carry=$c', round(d['value']), round(d['ms_per_step'],4), 'frac', round(r['frac'],3), 'shipped', round(r['frac_shipped'],3), 'iso', round(r['frac_isolated'],3), r['us_in_step'], r['us_in_step_deferred'])"
done
echo "== plan timelines"
VHAP_TEX_CARRY=0 timeout 300 python tools/plan_timeline.py --out $O/plan_timeline_carry0.txt > $O/plan_timeline0.log 2>&1 ; echo rc=$?
VHAP_TEX_CARRY=1 timeout 300 python tools/plan_timeline.py --out $O/plan_timeline_carry1.txt > $O/plan_timeline1.log 2>&1 ; echo rc=$?
grep -E "untimed|host enqueue" $O/plan_timeline_carry0.txt $O/plan_timeline_carry1.txt
