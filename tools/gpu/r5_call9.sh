#!/bin/bash
# round 5, GPU call 9: timeline of the restructured sharded step (world-size-1 RCCL group, default hardware queues)
set +e
O=gpurun_out/r5c9
mkdir -p $O
R="$GRAFT_REPO_ROOT"
export PYTHONUNBUFFERED=1
cd /tmp && export TMPDIR=/tmp
VHAP_FORCE_DIST=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o step -- python $R/bench.py --steps 60 --warmup 20 --no-cpu-baseline --no-stage --no-parity > $R/$O/bench_forced_rocprof.json 2> $R/$O/rocprof.err ; echo rc=$?
cd "$R"
KT=$(ls $O/prof/*kernel_trace.csv 2>/dev/null | head -1)
[ -n "$KT" ] && python tools/step_profile.py $KT --timeline > $O/sharded_step_timeline.txt 2>&1
rm -rf $O/prof
cat $O/sharded_step_timeline.txt | cut -c1-100
