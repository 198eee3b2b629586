#!/bin/bash
# round 4, GPU call 26: occupancy bounds after the load batching (rasteriser mode 2 at 6 waves/SIMD; shading backward at 4 with spills) A/B
set +e
O=gpurun_out/r4c26
mkdir -p $O
cd "$GRAFT_REPO_ROOT"
export PYTHONUNBUFFERED=1
for v in main tapsseq sbw4 base main tapsseq sbw4 base; do
  lib=$PWD/vhap_amd/lib/libvhap_hip_$v.so; [ $v = main ] && lib=$PWD/vhap_amd/lib/libvhap_hip.so
  VHAP_HIP_LIB=$lib timeout 300 python bench.py --no-cpu-baseline --no-stage --no-parity > $O/bench_$v.json 2> $O/bench.err
  python -c "import json; d=json.load(open('$O/bench_$v.json')); r=d['roofline']; print('$v', round(d['ms_per_step'],4), round(d['value']), r.get('us_in_step_deferred'), r.get('frac'))" | tee -a $O/ab.txt
done
for v in main sbw4; do
  lib=$PWD/vhap_amd/lib/libvhap_hip_$v.so; [ $v = main ] && lib=$PWD/vhap_amd/lib/libvhap_hip.so
  VHAP_HIP_LIB=$lib timeout 300 python tools/plan_timeline.py --out $O/plan_timeline_$v.txt > /dev/null 2>&1; echo $v; grep -E "raster_kernel|deferred_shade_bwd|gbuffer_bwd|texgrad_tile|untimed" $O/plan_timeline_$v.txt | grep -v " s[0-9] " | cut -c1-150; done
