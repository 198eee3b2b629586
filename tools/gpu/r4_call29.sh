#!/bin/bash
# round 4, GPU call 29: disturbance apply at 4 pixels per thread, photometric sum at 8 pixels per trip (A/B against 1 / 4), the tests call 28 did not reach
set +e
O=gpurun_out/r4c29
mkdir -p $O
cd "$GRAFT_REPO_ROOT"
export PYTHONUNBUFFERED=1
timeout 1200 python -m pytest tests -m gpu -q -x -k "not ten_steps and not fullbatch and not two_rank" > $O/pytest.log 2>&1; echo rc=$?; tail -4 $O/pytest.log | cut -c1-250
for v in main photo4 apt1 main photo4 apt1; do
  lib=$PWD/vhap_amd/lib/libvhap_hip_$v.so; [ $v = main ] && lib=$PWD/vhap_amd/lib/libvhap_hip.so
  VHAP_HIP_LIB=$lib timeout 300 python bench.py --no-cpu-baseline --no-stage --no-parity > $O/bench_$v.json 2> $O/bench.err
  python -c "import json; d=json.load(open('$O/bench_$v.json')); r=d['roofline']; print('$v', round(d['ms_per_step'],4), round(d['value']), r.get('us_in_step_deferred'), r.get('frac'))" | tee -a $O/ab.txt
done
for v in main photo4 apt1; do
  lib=$PWD/vhap_amd/lib/libvhap_hip_$v.so; [ $v = main ] && lib=$PWD/vhap_amd/lib/libvhap_hip.so
  VHAP_HIP_LIB=$lib timeout 300 python tools/plan_timeline.py --out $O/plan_timeline_$v.txt > /dev/null 2>&1; echo $v; grep -E "untimed|photo_fwd|disturb_apply|aa_detect" $O/plan_timeline_$v.txt | grep -v " s[0-9] " | cut -c1-100; done
