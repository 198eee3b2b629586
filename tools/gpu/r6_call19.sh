#!/bin/bash
# round 6, GPU call 19: landmark detection on ROCm -- the convolution / network / detector tests, the network's timing
set +e
O=gpurun_out/r6c19
mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 1200 python -m pytest tests/test_landmarks.py -q -x 2>&1 | tail -15
timeout 600 python tools/fan_bench.py --torch 2>&1 | tail -3 | tee $O/r06_fan_bench.txt
