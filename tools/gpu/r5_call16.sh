#!/bin/bash
# round 5, GPU call 16: the parity tests whose signatures changed after the suite of call 14 (full-batch yardsticks of configs 3 / 4 off by default)
set +e
O=gpurun_out/r5c16
mkdir -p $O
cd "$GRAFT_REPO_ROOT"
export PYTHONUNBUFFERED=1
timeout 1500 python -m pytest tests/test_parity_fullbatch_gpu.py "tests/test_parity_sizes_gpu.py::test_shipped_native_step_with_injected_disturbance_config2_size" tests/test_perf_guard_gpu.py -m gpu -q --durations=5 > $O/pytest.log 2>&1; echo rc=$?; tail -12 $O/pytest.log | cut -c1-200
