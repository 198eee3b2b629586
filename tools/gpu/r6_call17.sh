#!/bin/bash
# round 6, GPU call 17: the 5-step trajectories at the three BASELINE batches, HIP vs the float64 oracle fit, with the float32-oracle yardstick
# run in the SAME invocation and the gates of the round-5 review (1e-3 or 1.5 x yardstick per exported array; update-relative 2e-2)
set +e
O=gpurun_out/r6c17
mkdir -p $O
export PYTHONUNBUFFERED=1
for c in 2 4 3; do
  timeout 1500 python tools/fullbatch_trajectory.py both --config $c --record $O/r06_trajectory_5steps_cfg$c.txt --threads 16 > $O/traj_cfg$c.log 2>&1
  echo "cfg$c rc=$?"
  tail -3 $O/r06_trajectory_5steps_cfg$c.txt
done
