#!/bin/bash
# round 4, GPU call 4: the raster kernel as two concurrent launches (compute blocks || background blocks) vs one launch
set +e
O=gpurun_out/r4c4
mkdir -p $O
cd "$GRAFT_REPO_ROOT"
export PYTHONUNBUFFERED=1
for i in 1 2; do python tools/split_probe.py 2>&1 | grep "one launch" | tee -a $O/split_probe.txt; done
