#!/bin/bash
# round 4, GPU call 16: the disturbance's count pass fused into the deferred rasteriser: parity tests, A/B of the step
set +e
O=gpurun_out/r4c16
mkdir -p $O
cd "$GRAFT_REPO_ROOT"
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_parity_sizes_gpu.py tests/test_fused_gpu.py tests/test_deferred_gpu.py tests/test_plan_gpu.py tests/test_native_gpu.py -m gpu -q -x -k "not ten_steps" > $O/pytest.log 2>&1; echo rc=$?; tail -6 $O/pytest.log | cut -c1-250
for v in 1 0 1 0; do
  VHAP_DISTURB_COUNTED=$v timeout 300 python bench.py --no-cpu-baseline --no-stage --no-parity > $O/bench_counted$v.json 2> $O/bench.err
  python -c "import json; d=json.load(open('$O/bench_counted$v.json')); r=d['roofline']; print('counted=$v', round(d['ms_per_step'],4), round(d['value']), r['us_in_step_deferred'])" | tee -a $O/ab.txt
done
timeout 300 python tools/plan_timeline.py --out $O/plan_timeline.txt > $O/plan_timeline.log 2>&1 ; grep -E "untimed" $O/plan_timeline.txt
