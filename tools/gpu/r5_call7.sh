#!/bin/bash
# round 5, GPU call 7: where the sharded step (world-size-1 RCCL group) loses 0.45 ms against the one-plan step: kernel trace + timeline + host time
set +e
O=gpurun_out/r5c7
mkdir -p $O
R="$GRAFT_REPO_ROOT"
cd /tmp && export TMPDIR=/tmp
export PYTHONUNBUFFERED=1
VHAP_FORCE_DIST=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o step -- python $R/bench.py --steps 60 --warmup 20 --no-cpu-baseline --no-stage --no-parity > $R/$O/bench_forced_rocprof.json 2> $R/$O/rocprof.err ; echo rc=$?
cd "$R"
KT=$(ls $O/prof/*kernel_trace.csv 2>/dev/null | head -1)
[ -n "$KT" ] && python tools/step_profile.py $KT > $O/sharded_step_per_kernel.txt 2>&1 && python tools/step_profile.py $KT --timeline > $O/sharded_step_timeline.txt 2>&1
rm -rf $O/prof
head -50 $O/sharded_step_per_kernel.txt | cut -c1-150
wc -l $O/sharded_step_timeline.txt
python - <<'PY'
import json, os, subprocess, sys, time
# host time of one sharded replay: enqueue 40 replays without synchronising in between (the queue permitting) and divide
os.environ["VHAP_FORCE_DIST"] = "1"
sys.path.insert(0, os.getcwd())
import importlib.util, torch
spec = importlib.util.spec_from_file_location("bench", "bench.py"); bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
from vhap_amd import dist as vdist
from vhap_amd.tracker import GraphedStep
vdist.init_from_env(None)
C = bench.CONFIGS[2]
tr, own, n_local, model, topo, gt = bench.build_tracker(C, 0, 1, "cuda:0", "weak")
vdist.attach(tr)
opt = tr.configure_optimizer(tr.get_train_parameters(bench.STAGE), lr_scale=0.1)
st = GraphedStep(tr, tr.get_sample(own, device_index=True), opt, bench.STAGE)
with st.replay_stream():
    for _ in range(10): st()
    torch.cuda.synchronize()
    for n in (8, 8):
        t0 = time.perf_counter()
        for _ in range(n): st()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"sharded: host enqueue {1e6*(t1-t0)/n:.0f} us per step, incl. drain {1e6*(t2-t0)/n:.0f} us per step")
torch.distributed.destroy_process_group()
PY
