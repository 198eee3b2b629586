#!/bin/bash
# round 5, GPU call 19: BASELINE config 1 on the GPU (single 256 x 256 frame, mean shape, landmark-only) against the oracle fit
set +e
O=gpurun_out/r5c19
mkdir -p $O
cd "$GRAFT_REPO_ROOT"
export PYTHONUNBUFFERED=1
timeout 170 python -m pytest tests/test_config1_gpu.py -m gpu -q > $O/pytest.log 2>&1 ; echo rc=$?
tail -25 $O/pytest.log | cut -c1-260
