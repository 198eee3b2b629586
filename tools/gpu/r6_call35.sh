#!/bin/bash
# round 6, call 35: the list-driven shading backward -- the tests that replay or compare the step
set +e
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 2400 python -m pytest tests/test_deferred_gpu.py tests/test_native_gpu.py tests/test_fullsize_gpu.py tests/test_tex_carry_gpu.py tests/test_plan_gpu.py tests/test_energy_gpu.py tests/test_parity_sizes_gpu.py tests/test_fit_parity_gpu.py tests/test_perf_guard_gpu.py -q 2>&1 | tail -15
