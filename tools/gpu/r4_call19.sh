#!/bin/bash
# round 4, GPU call 19: read-before-write hunt (VHAP_POISON=1: uninitialised buffers start as NaN / 0x7f7f7f7f)
set +e
O=gpurun_out/r4c19
mkdir -p $O
cd "$GRAFT_REPO_ROOT"
export PYTHONUNBUFFERED=1
VHAP_POISON=1 timeout 900 python -m pytest tests/test_plan_gpu.py tests/test_parity_sizes_gpu.py -m gpu -q -k "multiview or config4 or config2_size" > $O/pytest_poison.log 2>&1; echo rc=$?; tail -12 $O/pytest_poison.log | cut -c1-300
cat gpurun_out/plan_stage_loop_multiview_feed.txt 2>/dev/null | head -14
