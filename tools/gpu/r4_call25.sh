#!/bin/bash
# round 4, GPU call 25: shading backward on 8x8 blocks per wave (vs 64 pixels of a row); rasteriser taps batched vs sequential: parity + A/B
set +e
O=gpurun_out/r4c25
mkdir -p $O
cd "$GRAFT_REPO_ROOT"
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_parity_sizes_gpu.py tests/test_deferred_gpu.py tests/test_plan_gpu.py tests/test_native_gpu.py tests/test_fullsize_gpu.py -m gpu -q -x -k "not ten_steps" > $O/pytest.log 2>&1; echo rc=$?; tail -3 $O/pytest.log | cut -c1-250
for v in main tapsseq sblinear base main tapsseq sblinear base; do
  lib=$PWD/vhap_amd/lib/libvhap_hip_$v.so; [ $v = main ] && lib=$PWD/vhap_amd/lib/libvhap_hip.so
  VHAP_HIP_LIB=$lib timeout 300 python bench.py --no-cpu-baseline --no-stage --no-parity > $O/bench_$v.json 2> $O/bench.err
  python -c "import json; d=json.load(open('$O/bench_$v.json')); r=d['roofline']; print('$v', round(d['ms_per_step'],4), round(d['value']), r.get('us_in_step_deferred'), r.get('frac'))" | tee -a $O/ab.txt
done
for v in main tapsseq; do
  lib=$PWD/vhap_amd/lib/libvhap_hip_$v.so; [ $v = main ] && lib=$PWD/vhap_amd/lib/libvhap_hip.so
  VHAP_HIP_LIB=$lib timeout 300 python tools/plan_timeline.py --out $O/plan_timeline_$v.txt > /dev/null 2>&1; echo $v; grep -E "raster_kernel|deferred_shade_bwd|gbuffer_bwd|texgrad_tile|untimed" $O/plan_timeline_$v.txt | grep -v " s[0-9] " | cut -c1-150; done
