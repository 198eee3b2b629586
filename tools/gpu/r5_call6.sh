#!/bin/bash
# round 5, GPU call 6: the reworked gradient gate at every BASELINE batch, the 5-step full-batch trajectories (HIP + oracle in one process,
# three configs side by side on the host cores), the remaining spread variants, the sharded step on a world-size-1 RCCL group
set +e
O=gpurun_out/r5c6
mkdir -p $O
cd "$GRAFT_REPO_ROOT"
export PYTHONUNBUFFERED=1
for c in 2 3 4; do
  (OMP_NUM_THREADS=16 timeout 1500 python tools/fullbatch_trajectory.py both --config $c --threads 16 --record $O/trajectory_cfg$c.txt > $O/trajectory_cfg$c.log 2>&1; echo "trajectory cfg$c rc=$?") &
done
OMP_NUM_THREADS=32 timeout 1800 python -m pytest tests/test_parity_fullbatch_gpu.py tests/test_parity_sizes_gpu.py -m gpu -q > $O/pytest_parity.log 2>&1; echo rc_parity=$?; tail -12 $O/pytest_parity.log | cut -c1-400
grep -h "spread\|L1 residuals" gpurun_out/parity_fullbatch_cfg*.txt gpurun_out/parity_native_injected_cfg*.txt | cut -c1-200
cp gpurun_out/parity_fullbatch_cfg*.txt gpurun_out/parity_native_injected_cfg*.txt gpurun_out/fit_parity_10_steps_512_T2048.txt $O/ 2>/dev/null
timeout 600 python tools/grad_spread_terms.py > $O/grad_spread_terms.txt 2> $O/grad_spread_terms.err; grep -c ratio $O/grad_spread_terms.txt
VHAP_FORCE_DIST=1 timeout 300 python bench.py --steps 100 --no-cpu-baseline --no-stage --no-parity > $O/bench_forced_dist.json 2> $O/bench_forced_dist.err
timeout 300 python bench.py --steps 100 --no-cpu-baseline --no-stage --no-parity > $O/bench_plain.json 2> $O/bench_plain.err
python -c "
import json
for n in ('forced_dist','plain'):
    d=json.load(open('$O/bench_'+n+'.json')); print(n, round(d['ms_per_step'],4), d['config']['parallelism'][:80], d['config'].get('tex_sharded'))
"
wait
tail -4 $O/trajectory_cfg*.txt | cut -c1-200
du -sh gpurun_out
