#!/bin/bash
# round 4, GPU call 2: A/B of raster-kernel residency variants (SGPR cap, min waves), diagnosis of the full-batch texture-gradient distance
set +e
O=gpurun_out/r4c2
mkdir -p $O
R="$GRAFT_REPO_ROOT"
cd "$R"
export PYTHONUNBUFFERED=1
echo "== A/B isolated (rocprofv3 kernel stats)"
bash tools/ab_libs.sh base s80 s80w8 2>&1 | tee $O/ab_isolated.txt
echo "== A/B in the step"
for v in base s80 s80w8; do
  VHAP_HIP_LIB=$R/vhap_amd/lib/libvhap_hip_$v.so timeout 300 python bench.py --no-cpu-baseline --no-stage --no-parity > $O/bench_$v.json 2> $O/bench_$v.err
  python -c "import json; d=json.load(open('$O/bench_$v.json')); r=d['roofline']; print('$v', round(d['ms_per_step'],4), round(r['frac'],4), round(r['frac_in_step_deferred'],4), round(r['frac_isolated'],4), r['us_in_step'], r['us_in_step_deferred'])" | tee -a $O/ab_in_step.txt
done
echo "== texture-gradient diagnosis"
timeout 900 python tools/diag_texgrad.py cfg2 > $O/diag_texgrad_cfg2.txt 2>&1; cat $O/diag_texgrad_cfg2.txt | tail -12
