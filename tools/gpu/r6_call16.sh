#!/bin/bash
# round 6, GPU call 16: reproduce the full-suite failure of test_c_abi_is_reentrant_across_threads: every GPU test file that precedes it, one process
set +e
export PYTHONUNBUFFERED=1
timeout 1500 python -m pytest tests/test_config1_gpu.py tests/test_deferred_gpu.py tests/test_dist_gpu.py tests/test_energy_gpu.py tests/test_fit_parity_gpu.py tests/test_fullsize_gpu.py tests/test_fused_gpu.py tests/test_ingest.py tests/test_native_gpu.py -m gpu -q -x 2>&1 | grep -v "Warning\|warn\|^  " | tail -30
echo "== rccl one rank"
timeout 900 python -m pytest "tests/test_dist_gpu.py::test_one_rank_rccl_sharded_step_matches_unsharded" -q 2>&1 | grep -v "^  \|Warning\|warn" | grep -B2 -A25 "Error\|assert" | head -60
