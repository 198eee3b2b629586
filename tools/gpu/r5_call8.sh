#!/bin/bash
# round 5, GPU call 8: the sharded step with the texture chain as the pixel plan's open tail + communication stream; hardware queues 4 vs 8
set +e
O=gpurun_out/r5c8
mkdir -p $O
R="$GRAFT_REPO_ROOT"
cd "$R"
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_dist_gpu.py -m gpu -q -x > $O/pytest_dist.log 2>&1; echo rc_dist=$?; tail -6 $O/pytest_dist.log | cut -c1-300
for q in 4 8; do
  GPU_MAX_HW_QUEUES=$q VHAP_FORCE_DIST=1 timeout 300 python bench.py --steps 100 --no-cpu-baseline --no-stage --no-parity > $O/bench_forced_q$q.json 2> $O/bench_forced_q$q.err
  GPU_MAX_HW_QUEUES=$q timeout 300 python bench.py --steps 100 --no-cpu-baseline --no-stage --no-parity > $O/bench_plain_q$q.json 2> $O/bench_plain_q$q.err
  python -c "
import json
for n in ('forced','plain'):
    d=json.load(open('$O/bench_%s_q$q.json' % n)); print('queues $q', n, round(d['ms_per_step'],4), 'frac', round(d['roofline']['frac'],3) if d['roofline']['frac'] else None)
" | tee -a $O/summary.txt
done
cd /tmp && export TMPDIR=/tmp
GPU_MAX_HW_QUEUES=8 VHAP_FORCE_DIST=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o step -- python $R/bench.py --steps 60 --warmup 20 --no-cpu-baseline --no-stage --no-parity > $R/$O/bench_forced_rocprof.json 2> $R/$O/rocprof.err ; echo rc=$?
cd "$R"
KT=$(ls $O/prof/*kernel_trace.csv 2>/dev/null | head -1)
[ -n "$KT" ] && python tools/step_profile.py $KT --timeline > $O/sharded_step_timeline.txt 2>&1
rm -rf $O/prof
head -3 $O/sharded_step_timeline.txt; grep -n "oneRank\|gbuffer_bwd\|texgrad\|mip_fold\|copyBuffer\|tex_prep_bwd" $O/sharded_step_timeline.txt | cut -c1-110
