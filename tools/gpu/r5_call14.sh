#!/bin/bash
# round 5, GPU call 14: the whole GPU suite on the round's tree (first pass: find what breaks), the fp32-oracle yardstick of the config-3 trajectory
# on the host cores meanwhile, smoke
set +e
O=gpurun_out/r5c14
mkdir -p $O
R="$GRAFT_REPO_ROOT"
cd "$R"
export PYTHONUNBUFFERED=1
( OMP_NUM_THREADS=16 timeout 300 python tools/fullbatch_trajectory.py gpu --config 3 --out /tmp/traj_cfg3.npz > $O/yardstick_gpu.log 2>&1
  ( OMP_NUM_THREADS=16 timeout 1500 python tools/fullbatch_trajectory.py cpu /tmp/traj_cfg3.npz --threads 16 --dtype float64 --save /tmp/exp64.npz > $O/yardstick_f64.log 2>&1 ) &
  ( OMP_NUM_THREADS=16 timeout 1500 python tools/fullbatch_trajectory.py cpu /tmp/traj_cfg3.npz --threads 16 --dtype float32 --save /tmp/exp32.npz > $O/yardstick_f32.log 2>&1 ) &
  wait
  python tools/fullbatch_trajectory.py yardstick /tmp/traj_cfg3.npz /tmp/exp64.npz /tmp/exp32.npz --record $O/trajectory_cfg3_fp32_yardstick.txt > /dev/null 2>&1; echo yardstick rc=$? ) &
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1; echo smoke rc=$?
timeout 3000 python -m pytest tests -m gpu -q --durations=12 > $O/pytest_gpu.log 2>&1 ; echo rc_suite=$?
tail -30 $O/pytest_gpu.log | cut -c1-300; grep -n "^E  " $O/pytest_gpu.log | head -20
wait
cat $O/trajectory_cfg3_fp32_yardstick.txt | cut -c1-220
rm -f gpurun_out/*.npz; du -sh gpurun_out
