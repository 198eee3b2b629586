#!/bin/bash
# round 3, GPU call 10: texture branch forked ahead of the camera, one hand-over for pair list + statistics; A/B occupancy of the shading backward
set +e
O=gpurun_out/r3c10
mkdir -p $O
R="$GRAFT_REPO_ROOT"
cd "$R"
export PYTHONUNBUFFERED=1
echo "== plan timeline: default / shading backward at 5 waves per SIMD (debug flag 131072)"
timeout 300 python tools/plan_timeline.py --out $O/plan_timeline.txt > $O/plan_timeline.log 2>&1 ; echo rc=$?
grep -E "untimed|host enqueue" $O/plan_timeline.txt; grep -E "deferred_shade_bwd|raster_kernel|tex_prep" $O/plan_timeline.txt | tail -4
timeout 300 python tools/plan_timeline.py --debug-flags 131072 --out $O/plan_timeline_w5.txt > $O/plan_timeline_w5.log 2>&1 ; echo rc=$?
grep -E "untimed|host enqueue" $O/plan_timeline_w5.txt; grep -E "deferred_shade_bwd" $O/plan_timeline_w5.txt | tail -1
echo "== tests touched by the change"
timeout 900 python -m pytest tests/test_plan_gpu.py tests/test_parity_sizes_gpu.py tests/test_energy_gpu.py tests/test_deferred_gpu.py tests/test_native_gpu.py tests/test_dist_gpu.py -q > $O/pytest_some.log 2>&1; tail -6 $O/pytest_some.log; grep -n "^E  " $O/pytest_some.log | head -20
