#!/bin/bash
# round 4, GPU call 21: uv / uv derivatives stored by the deferred rasteriser, shading backward with one round trip less: parity + A/B
set +e
O=gpurun_out/r4c21
mkdir -p $O
cd "$GRAFT_REPO_ROOT"
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_parity_sizes_gpu.py tests/test_deferred_gpu.py tests/test_plan_gpu.py tests/test_native_gpu.py tests/test_fullsize_gpu.py -m gpu -q -x -k "not ten_steps" > $O/pytest.log 2>&1; echo rc=$?; tail -6 $O/pytest.log | cut -c1-250
for v in 1 0 1 0; do
  VHAP_UV_STORED=$v timeout 300 python bench.py --no-cpu-baseline --no-stage --no-parity > $O/bench_uv$v.json 2> $O/bench.err
  python -c "import json; d=json.load(open('$O/bench_uv$v.json')); r=d['roofline']; print('uv_stored=$v', round(d['ms_per_step'],4), round(d['value']), r['us_in_step_deferred'])" | tee -a $O/ab.txt
done
for v in 1 0; do VHAP_UV_STORED=$v timeout 300 python tools/plan_timeline.py --out $O/plan_timeline_uv$v.txt > /dev/null 2>&1; grep -E "raster_kernel|deferred_shade_bwd|untimed" $O/plan_timeline_uv$v.txt | cut -c1-150; done
