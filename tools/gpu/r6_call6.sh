#!/bin/bash
# round 6, GPU call 6: the whole GPU suite on the carried-texture tree (no -x: every failure in one go)
set +e
O=gpurun_out/r6c6
mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -40 | tee $O/pytest_gpu.log
