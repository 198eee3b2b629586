#!/bin/bash
# round 4, GPU call 28: camera forward / backward as workgroups of the per-frame launches (ABI 8), frame_prep_fwd with two load levels, aa_detect batched:
# all GPU tests but the long ones + A/B of the camera fusion + timeline + kernel trace
set +e
O=gpurun_out/r4c28
mkdir -p $O
cd "$GRAFT_REPO_ROOT"
export PYTHONUNBUFFERED=1
timeout 1200 python -m pytest tests -m gpu -q -x -k "not ten_steps and not fullbatch and not two_rank" > $O/pytest.log 2>&1; echo rc=$?; tail -4 $O/pytest.log | cut -c1-250
for v in 1 0 1 0; do
  VHAP_CAMERA_FUSED=$v timeout 300 python bench.py --no-cpu-baseline --no-stage --no-parity > $O/bench_cam$v.json 2> $O/bench.err
  python -c "import json; d=json.load(open('$O/bench_cam$v.json')); r=d['roofline']; print('camera_fused=$v', round(d['ms_per_step'],4), round(d['value']), r.get('us_in_step_deferred'), r.get('frac'))" | tee -a $O/ab.txt
done
timeout 300 python tools/plan_timeline.py --out $O/plan_timeline.txt > /dev/null 2>&1
grep -E "untimed" $O/plan_timeline.txt
sed -n '/timed replays/,$p' $O/plan_timeline.txt | cut -c1-110
