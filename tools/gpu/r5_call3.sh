#!/bin/bash
# round 5, GPU call 3 (early stores v2: box from a launch of its own, row-run stores): early stores (VHAP_RASTER_PREFILL) -- bit-exactness, then what they buy: isolated RI-fwd, the step with / without
set +e
O=gpurun_out/r5c3
mkdir -p $O
cd "$GRAFT_REPO_ROOT"
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_raster_gpu.py tests/test_deferred_gpu.py -m gpu -q -x > $O/pytest.log 2>&1; echo rc=$?; tail -12 $O/pytest.log | cut -c1-300
for pf in 1 0; do
  VHAP_PREFILL=$pf VHAP_DEBUG=$([ $pf = 0 ] && echo 8192 || echo 0) timeout 400 python bench.py --steps 100 --no-cpu-baseline --no-stage --no-parity > $O/bench_pf$pf.json 2> $O/bench_pf$pf.err
  python -c "import json; d=json.load(open('$O/bench_pf$pf.json')); r=d['roofline']; print('prefill $pf', round(d['ms_per_step'],4), 'frac', round(r['frac'],3), 'iso', round(r['frac_isolated'],3), 'deferred', round(r['frac_in_step_deferred'],3), r['us_in_step'], r['us_in_step_deferred'], 'defer_join', d['config']['deferred_join'])" | tee -a $O/summary.txt
done
tail -3 $O/bench_pf1.err
