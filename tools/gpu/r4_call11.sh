#!/bin/bash
# round 4, GPU call 11: buffer-based deferred-join decision, node -> call mapping by node handle
set +e
O=gpurun_out/r4c11
mkdir -p $O
cd "$GRAFT_REPO_ROOT"
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_plan_gpu.py -m gpu -q -x > $O/pytest_plan.log 2>&1; echo rc=$?; tail -40 $O/pytest_plan.log | cut -c1-300
cp gpurun_out/plan_deferred_join*.txt $O/ 2>/dev/null
