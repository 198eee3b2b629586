#!/bin/bash
# round 3, GPU call 15: same-box A/B of {general, pipelined} finish kernel x {normal, low} side-stream priority
set +e
O=gpurun_out/r3c15
mkdir -p $O
R="$GRAFT_REPO_ROOT"
cd "$R"
export PYTHONUNBUFFERED=1
for f in 262144 786432 524288 0 786432 262144; do
  timeout 300 python tools/plan_timeline.py --reps 3 --debug-flags $f --out $O/plan_timeline_$f.txt > $O/plan_timeline_$f.log 2>&1 ; echo flags=$f rc=$?
  grep -E "untimed" $O/plan_timeline_$f.txt
done
