#!/bin/bash
# round 6, call 43: the config-3 trajectory again (the energy of the later steps follows the parameters: a note when every array gate holds)
set +e
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp PYTHONUNBUFFERED=1
O=gpurun_out/r6c43; mkdir -p $O
timeout 1200 python tools/fullbatch_trajectory.py both --config 3 --record $O/r06_trajectory_5steps_cfg3.txt --threads 16 > $O/traj_cfg3.log 2>&1 ; echo "cfg3 rc=$?"
tail -6 $O/r06_trajectory_5steps_cfg3.txt
