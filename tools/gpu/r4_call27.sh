#!/bin/bash
# round 4, GPU call 27: loads batched in the latency-bound kernels of the main chain (frame_prep, skinning, vertex stage, disturbance, photometric sum,
# uv binning, tile accumulation): all GPU tests but the long ones + A/B against the previous commit
set +e
O=gpurun_out/r4c27
mkdir -p $O
cd "$GRAFT_REPO_ROOT"
export PYTHONUNBUFFERED=1
timeout 1200 python -m pytest tests -m gpu -q -x -k "not ten_steps and not fullbatch and not two_rank" > $O/pytest.log 2>&1; echo rc=$?; tail -4 $O/pytest.log | cut -c1-250
for v in main base main base; do
  lib=$PWD/vhap_amd/lib/libvhap_hip_$v.so; [ $v = main ] && lib=$PWD/vhap_amd/lib/libvhap_hip.so
  VHAP_HIP_LIB=$lib timeout 300 python bench.py --no-cpu-baseline --no-stage --no-parity > $O/bench_$v.json 2> $O/bench.err
  python -c "import json; d=json.load(open('$O/bench_$v.json')); r=d['roofline']; print('$v', round(d['ms_per_step'],4), round(d['value']), r.get('us_in_step_deferred'), r.get('frac'))" | tee -a $O/ab.txt
done
for v in main base; do
  lib=$PWD/vhap_amd/lib/libvhap_hip_$v.so; [ $v = main ] && lib=$PWD/vhap_amd/lib/libvhap_hip.so
  VHAP_HIP_LIB=$lib timeout 300 python tools/plan_timeline.py --out $O/plan_timeline_$v.txt > /dev/null 2>&1; done
grep -E "untimed" $O/plan_timeline_main.txt $O/plan_timeline_base.txt
python - <<P
import re
def rows(f):
    d={}
    for l in open(f):
        m=re.match(r'\s+([\d.]+)\s+([\d.]+)\s+n(\d+)\s+(\S+)', l)
        if m: d[int(m.group(3))]=(float(m.group(1)),float(m.group(2)),m.group(4)[:60])
    return d
a=rows('$O/plan_timeline_main.txt'); b=rows('$O/plan_timeline_base.txt')
for k in sorted(a, key=lambda k:a[k][0]):
    if k in b: print(f"n{k:<3} {a[k][2][18:58]:40s} start {a[k][0]:7.1f} ({b[k][0]:7.1f})  dur {a[k][1]:6.1f} ({b[k][1]:6.1f})")
P
