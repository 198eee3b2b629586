#!/bin/bash
# round 4, GPU call 24: loads batched by dependency level in the pixel kernels (closed-form mip offsets, 8-tap fetch, stand-in addresses): parity + A/B
set +e
O=gpurun_out/r4c24
mkdir -p $O
cd "$GRAFT_REPO_ROOT"
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_parity_sizes_gpu.py tests/test_deferred_gpu.py tests/test_plan_gpu.py tests/test_native_gpu.py tests/test_fullsize_gpu.py tests/test_ops_gpu.py tests/test_fused_gpu.py tests/test_raster_gpu.py -m gpu -q -x -k "not ten_steps" > $O/pytest.log 2>&1; echo rc=$?; tail -6 $O/pytest.log | cut -c1-250
for v in new base new base; do
  lib=$PWD/vhap_amd/lib/libvhap_hip.so; [ $v = base ] && lib=$PWD/vhap_amd/lib/libvhap_hip_base.so
  VHAP_HIP_LIB=$lib timeout 300 python bench.py --no-cpu-baseline --no-stage --no-parity > $O/bench_$v.json 2> $O/bench.err
  python -c "import json; d=json.load(open('$O/bench_$v.json')); r=d['roofline']; print('$v', round(d['ms_per_step'],4), round(d['value']), r.get('us_in_step_deferred'), r.get('frac'))" | tee -a $O/ab.txt
done
for v in new base; do
  lib=$PWD/vhap_amd/lib/libvhap_hip.so; [ $v = base ] && lib=$PWD/vhap_amd/lib/libvhap_hip_base.so
  VHAP_HIP_LIB=$lib timeout 300 python tools/plan_timeline.py --out $O/plan_timeline_$v.txt > /dev/null 2>&1; echo $v; grep -E "raster_kernel|deferred_shade_bwd|gbuffer_bwd|texgrad_tile|untimed" $O/plan_timeline_$v.txt | cut -c1-150; done
