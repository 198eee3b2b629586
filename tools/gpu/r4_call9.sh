#!/bin/bash
# round 4, GPU call 9: stage loop after (all epochs drawn ahead, one upload, lr table by value kernel); plan + native tests
set +e
O=gpurun_out/r4c9
mkdir -p $O
cd "$GRAFT_REPO_ROOT"
export PYTHONUNBUFFERED=1
timeout 300 python tools/stage_timeline.py --frames 256 --epochs 5 --out $O/stage_timeline_after.txt 2>&1 | tail -9
timeout 300 python tools/stage_timeline.py --frames 64 --epochs 3 --out $O/stage_timeline_after_64x3.txt 2>&1 | tail -9
timeout 900 python -m pytest tests/test_plan_gpu.py tests/test_native_gpu.py tests/test_energy_gpu.py -m gpu -q -x > $O/pytest.log 2>&1; echo rc=$?; tail -5 $O/pytest.log
cp gpurun_out/plan_deferred_join_stage_loop.txt $O/ 2>/dev/null
