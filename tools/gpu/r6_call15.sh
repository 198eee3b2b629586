#!/bin/bash
# round 6, GPU call 15: which earlier test file makes test_c_abi_is_reentrant_across_threads fail in a full-suite run (bisect by file groups);
# the one-rank RCCL tests with the fills recorded
set +e
export PYTHONUNBUFFERED=1
T=tests/test_native_gpu.py::test_c_abi_is_reentrant_across_threads
for grp in "tests/test_config1_gpu.py tests/test_deferred_gpu.py" "tests/test_energy_gpu.py tests/test_fit_parity_gpu.py" "tests/test_fullsize_gpu.py tests/test_fused_gpu.py tests/test_ingest.py" "tests/test_dist_gpu.py"; do
  echo "== $grp"
  timeout 900 python -m pytest $grp $T -q -m gpu 2>&1 | tail -4
done
echo "== rccl one rank"
timeout 900 python -m pytest "tests/test_dist_gpu.py::test_one_rank_rccl_sharded_step_matches_unsharded" -q 2>&1 | grep -v "^  \|Warning\|warn" | tail -12
