#!/bin/bash
# round 3, GPU call 14: software-pipelined texture assembly and texture finish + Adam; A/B: general finish kernel (262144), low-priority side streams (524288)
set +e
O=gpurun_out/r3c14
mkdir -p $O
R="$GRAFT_REPO_ROOT"
cd "$R"
export PYTHONUNBUFFERED=1
echo "== tests of the texture passes first"
timeout 900 python -m pytest tests/test_native_gpu.py tests/test_energy_gpu.py tests/test_fit_parity_gpu.py tests/test_plan_gpu.py -q > $O/pytest_some.log 2>&1; tail -6 $O/pytest_some.log; grep -n "^E  " $O/pytest_some.log | head -20
echo "== kbench (un-contended)"
timeout 300 python tools/kbench.py --only tex_forward,tex_finish > $O/kbench.txt 2>&1 ; tail -5 $O/kbench.txt
echo "== plan timeline: default / general finish kernel / low-priority side streams"
for f in 0 262144 524288; do
  timeout 300 python tools/plan_timeline.py --debug-flags $f --out $O/plan_timeline_$f.txt > $O/plan_timeline_$f.log 2>&1 ; echo flags=$f rc=$?
  grep -E "untimed" $O/plan_timeline_$f.txt; grep -E "tex_prep_fwd|tex_finish_adam|tex_prep_bwd|raster_kernel|frame_prep_fwd|flame_skin_fwd|bin_build" $O/plan_timeline_$f.txt | tail -7 | cut -c1-90
done
