#!/bin/bash
# round 6, GPU call 7: the tests that failed in call 6 (mid-edit snapshot, tex_extra.grad) + the dist file with its new tests; bench with the
# background d_albedo stores skipped
set +e
O=gpurun_out/r6c7
mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 1500 python -m pytest tests/test_native_gpu.py::test_c_abi_is_reentrant_across_threads tests/test_energy_gpu.py::test_graphed_step_matches_eager_step tests/test_tex_carry_gpu.py tests/test_deferred_gpu.py tests/test_perf_guard_gpu.py -q 2>&1 | tail -30
timeout 1500 python -m pytest tests/test_dist_gpu.py -q 2>&1 | tail -30
timeout 1500 python -m pytest tests/test_parity_sizes_gpu.py tests/test_parity_fullbatch_gpu.py -q 2>&1 | tail -12
for c in 1 1; do
  VHAP_TEX_CARRY=$c timeout 300 python bench.py --no-cpu-baseline --no-stage --no-parity > $O/bench$c.json 2> $O/bench$c.err
  python -c "
import json
d=json.load(open('$O/bench$c.json')); r=d['roofline']
print('carry=$c', round(d['value']), round(d['ms_per_step'],4), 'frac', round(r['frac'],3), 'shipped', round(r['frac_shipped'],3), 'iso', round(r['frac_isolated'],3), r['us_in_step'])"
done
ls gpurun_out/*.txt 2>/dev/null | head -30
