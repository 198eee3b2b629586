#!/bin/bash
# round 6, call 21: the one-rank RCCL test [False-False] inside its own file, four times (it misses its bound inside the suite, never alone)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r6c21; mkdir -p $O
for i in 1 2 3 4; do
  echo "== run $i"
  timeout 900 python -m pytest tests/test_dist_gpu.py -q 2>&1 | tail -4
  cp gpurun_out/dist_one_rank_rccl_00.txt $O/rccl_00_run$i.txt
  grep -n "off by\|of the update (floor" $O/rccl_00_run$i.txt
done
