#!/bin/bash
# round 6, call 20: the landmark tests again (the conv test held temporaries' pointers); why the one-rank RCCL test [False-False] misses its bound now and then
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r6c20; mkdir -p $O
echo "== landmarks"
timeout 900 python -m pytest tests/test_landmarks.py -q 2>&1 | tail -15
echo "== probe, snapshots of every step"
timeout 900 python tools/rccl_flake_probe.py --trials 7 > $O/probe_sync.txt 2>&1; echo rc=$?
grep -c "OFF" $O/probe_sync.txt; grep -n "OFF\|trial" $O/probe_sync.txt | head -60
echo "== probe, the test's own timing"
timeout 900 python tools/rccl_flake_probe.py --trials 7 --no-sync-each > $O/probe_nosync.txt 2>&1; echo rc=$?
grep -c "OFF" $O/probe_nosync.txt; grep -n "OFF\|trial" $O/probe_nosync.txt | head -60
