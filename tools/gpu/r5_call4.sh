#!/bin/bash
# round 5, GPU call 4: the new tests (ingest preparation, RNG distribution, early-stores parity in the step, perf guards), the gradient gate
# measured per parameter against the fp32 oracle at every BASELINE batch, the trajectory dumps, the background-load A/B, the thread sweep
set +e
O=gpurun_out/r5c4
mkdir -p $O
cd "$GRAFT_REPO_ROOT"
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_ingest.py tests/test_fused_gpu.py tests/test_perf_guard_gpu.py tests/test_raster_gpu.py::test_early_stores_outside_the_geometry_box tests/test_deferred_gpu.py::test_deferred_step_with_early_stores_is_bit_identical -m gpu -q > $O/pytest_new.log 2>&1; echo rc_new=$?; tail -25 $O/pytest_new.log | cut -c1-400
for c in 2 3 4; do
  timeout 300 python tools/fullbatch_trajectory.py gpu --config $c --out gpurun_out/traj_cfg$c.npz 2>&1 | tail -2
done
ls -la gpurun_out/traj_cfg*.npz
timeout 2400 python -m pytest tests/test_parity_fullbatch_gpu.py tests/test_parity_sizes_gpu.py -m gpu -q > $O/pytest_parity.log 2>&1; echo rc_parity=$?; tail -30 $O/pytest_parity.log | cut -c1-600
grep -h "spread" gpurun_out/parity_fullbatch_cfg*.txt | cut -c1-200
for dbg in 0 16384; do
  VHAP_DEBUG=$dbg timeout 300 python bench.py --steps 100 --no-cpu-baseline --no-stage --no-parity > $O/bench_dbg$dbg.json 2> $O/bench_dbg$dbg.err
  python -c "import json; d=json.load(open('$O/bench_dbg$dbg.json')); r=d['roofline']; print('debug $dbg', round(d['ms_per_step'],4), 'frac', round(r['frac'],3), 'shipped', round(r['frac_shipped'],3), r['us_in_step_deferred'])" | tee -a $O/bg_early_ab.txt
done
timeout 400 python tools/cpu_baseline_sweep.py > $O/cpu_baseline_thread_sweep.txt 2>&1; tail -5 $O/cpu_baseline_thread_sweep.txt
