#!/bin/bash
# round 6, call 42: the 5-step trajectories against the float64 oracle fit (with the float32-oracle yardstick) on the final tree
set +e
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp PYTHONUNBUFFERED=1
O=gpurun_out/r6c42; mkdir -p $O
for c in 2 4 3; do
  timeout 1200 python tools/fullbatch_trajectory.py both --config $c --record $O/r06_trajectory_5steps_cfg$c.txt --threads 16 > $O/traj_cfg$c.log 2>&1 ; echo "cfg$c rc=$?"
  tail -3 $O/r06_trajectory_5steps_cfg$c.txt
done
