#!/bin/bash
# round 5, GPU call 5: where the config-3 / config-4 gradient spread comes from (records kept SMALL this time), the two small failing parity tests
set +e
O=gpurun_out/r5c5
mkdir -p $O
cd "$GRAFT_REPO_ROOT"
export PYTHONUNBUFFERED=1
timeout 600 python tools/grad_spread_terms.py > $O/grad_spread_terms.txt 2> $O/grad_spread_terms.err; tail -70 $O/grad_spread_terms.txt
timeout 900 python -m pytest "tests/test_parity_sizes_gpu.py::test_shipped_native_step_config3_size_static_offset_trained" "tests/test_parity_sizes_gpu.py::test_shipped_native_step_config4_size_calibrated_views" -m gpu -q > $O/pytest_small.log 2>&1; echo rc=$?
cat gpurun_out/parity_native_injected_cfg3.txt gpurun_out/parity_native_injected_cfg4.txt | cut -c1-220
