#!/bin/bash
# round 5, GPU call 18: the whole GPU suite on the final tree of the round
set +e
O=gpurun_out/r5c18
mkdir -p $O
cd "$GRAFT_REPO_ROOT"
export PYTHONUNBUFFERED=1
timeout 1500 python -m pytest tests -m gpu -q --durations=6 > $O/pytest_gpu.log 2>&1 ; echo rc_suite=$?
tail -14 $O/pytest_gpu.log | cut -c1-200; grep -n "^E  " $O/pytest_gpu.log | head -10
