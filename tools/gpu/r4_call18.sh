#!/bin/bash
# round 4, GPU call 18: multi-view stage loop, self-feeding step vs host-fed
set +e
O=gpurun_out/r4c18
mkdir -p $O
cd "$GRAFT_REPO_ROOT"
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_plan_gpu.py -m gpu -q -x -k multiview > $O/pytest.log 2>&1; echo rc=$?; tail -25 $O/pytest.log | cut -c1-250
cp gpurun_out/plan_stage_loop_multiview_feed.txt $O/ 2>/dev/null; cat $O/plan_stage_loop_multiview_feed.txt 2>/dev/null
