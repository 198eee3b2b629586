#!/bin/bash
# round 6, call 23: split-K convolutions (tests, timing with / without, batch 2 and 16); the dist tests with the kink-aware repetition, 3 x
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r6c23; mkdir -p $O
echo "== landmarks"
timeout 900 python -m pytest tests/test_landmarks.py -q 2>&1 | tail -8
for b in 2 16; do
  timeout 600 python tools/fan_bench.py --batch $b --torch 2>&1 | tail -2 | tee -a $O/r06_fan_bench.txt
  timeout 600 python tools/fan_bench.py --batch $b --no-split 2>&1 | tail -1 | tee -a $O/r06_fan_bench.txt
done
echo "== rocprof of the network, batch 2"
cd /tmp; timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/fan_prof -o fan -- python $GRAFT_REPO_ROOT/tools/fan_bench.py --batch 2 --reps 20 > /dev/null 2>&1; cd $GRAFT_REPO_ROOT
f=$(ls $O/fan_prof/*/fan_kernel_stats.csv 2>/dev/null | head -1); head -12 "$f" | cut -c1-200; cp "$f" $O/fan_kernel_stats.csv; rm -rf $O/fan_prof
for i in 1 2 3; do
  echo "== dist run $i"
  timeout 900 python -m pytest tests/test_dist_gpu.py -q 2>&1 | tail -3
  grep -h "localised\|attempt [12]" gpurun_out/dist_one_rank_rccl_*.txt | cut -c1-200
done
