#!/bin/bash
# round 6, call 29: where the covered pixels sit in the order the backward pixel kernels walk
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp PYTHONUNBUFFERED=1
mkdir -p gpurun_out/r6c29
for c in 2 3 4; do timeout 300 python tools/coverage_stats.py --config $c 2>&1 | tail -6; done | tee gpurun_out/r6c29/coverage_stats.txt
