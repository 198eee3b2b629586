#!/bin/bash
# round 4, GPU call 17: batched per-frame offset regularisers + library offset combine (use_dynamic_offset)
set +e
O=gpurun_out/r4c17
mkdir -p $O
cd "$GRAFT_REPO_ROOT"
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_parity_sizes_gpu.py tests/test_energy_gpu.py tests/test_native_gpu.py -m gpu -q -x -k "dynamic or offset or native" > $O/pytest.log 2>&1; echo rc=$?; tail -6 $O/pytest.log | cut -c1-250
