#!/bin/bash
# round 4, GPU call 38: is the slow stage mode the cyclic collector?  Five processes, time in gc measured; two of them with the collector off
set +e
O=gpurun_out/r4c38
mkdir -p $O
cd "$GRAFT_REPO_ROOT"
export PYTHONUNBUFFERED=1
for i in 1 2 3 4 5; do
  g=1; [ $i -ge 4 ] && g=0
  STAGE_GC=$g timeout 100 python tools/stage_timeline.py --out $O/stage_$i.txt > /dev/null 2> $O/err_$i.txt; sed -n '2,3p' $O/stage_$i.txt | cut -c1-230; done
