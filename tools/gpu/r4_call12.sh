#!/bin/bash
# round 4, GPU call 12: sharded texture update (reduce-scatter -> row-strip finish + Adam -> all-gather), 2 ranks on one GPU over gloo
set +e
O=gpurun_out/r4c12
mkdir -p $O
cd "$GRAFT_REPO_ROOT"
export PYTHONUNBUFFERED=1
timeout 1200 python -m pytest tests/test_dist_gpu.py -m gpu -q -x > $O/pytest_dist.log 2>&1; echo rc=$?; tail -30 $O/pytest_dist.log | cut -c1-250
