#!/bin/bash
# round 4, GPU call 40: the stage with its objects frozen out of the collector's reach, three processes
set +e
O=gpurun_out/r4c40
mkdir -p $O
cd "$GRAFT_REPO_ROOT"
export PYTHONUNBUFFERED=1
for i in 1 2; do timeout 100 python tools/stage_timeline.py --out $O/stage_$i.txt > /dev/null 2> $O/err_$i.txt; sed -n '2,3p' $O/stage_$i.txt | cut -c1-230; done
