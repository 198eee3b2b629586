#!/bin/bash
# round 6, call 22: the convolution with its loads one step ahead (tests + timing); the one-rank RCCL comparison 40 x per variant with per-step snapshots
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r6c22; mkdir -p $O
echo "== landmarks"
timeout 900 python -m pytest tests/test_landmarks.py -q 2>&1 | tail -8
timeout 600 python tools/fan_bench.py --torch 2>&1 | tail -3 | tee $O/r06_fan_bench.txt
for v in "1 0" "0 0"; do
  set -- $v
  echo "== probe tex_sharded=$1 tex_first=$2"
  timeout 1200 python tools/rccl_flake_probe.py --trials 40 --tex-sharded $1 --tex-first $2 > $O/probe_$1$2.txt 2>&1; echo rc=$?
  grep -c "OFF" $O/probe_$1$2.txt; grep -n "OFF" $O/probe_$1$2.txt | cut -c1-230 | head -40
done
