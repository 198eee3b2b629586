#!/bin/bash
# round 4, GPU call 14: the self-feeding step (vhap_batch_feed as a plan node): plan tests, stage timeline, dist tests (gather fold)
set +e
O=gpurun_out/r4c14
mkdir -p $O
cd "$GRAFT_REPO_ROOT"
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_plan_gpu.py -m gpu -q -x > $O/pytest_plan.log 2>&1; echo rc=$?; tail -25 $O/pytest_plan.log | cut -c1-260
cp gpurun_out/plan_deferred_join*.txt $O/ 2>/dev/null
timeout 300 python tools/stage_timeline.py --frames 256 --epochs 5 --out $O/stage_timeline_feed.txt 2>&1 | tail -9
VHAP_STEP_FEED=0 timeout 300 python tools/stage_timeline.py --frames 256 --epochs 5 --out $O/stage_timeline_hostfed.txt 2>&1 | tail -9
timeout 900 python -m pytest tests/test_dist_gpu.py -m gpu -q -x > $O/pytest_dist.log 2>&1; echo rc=$?; tail -3 $O/pytest_dist.log | cut -c1-200
