#!/bin/bash
# round 4, GPU call 37: which mode is the slow stage in (deferred join / self-feeding as decided at capture)?  Two processes.
set +e
O=gpurun_out/r4c37
mkdir -p $O
cd "$GRAFT_REPO_ROOT"
export PYTHONUNBUFFERED=1
for i in 1 2; do timeout 200 python tools/stage_timeline.py --out $O/stage_$i.txt > /dev/null 2> $O/err_$i.txt; head -4 $O/stage_$i.txt | cut -c1-260; done
