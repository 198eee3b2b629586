#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp PYTHONUNBUFFERED=1
for i in 1 2 3; do timeout 600 python -m pytest tests/test_tex_carry_gpu.py -q 2>&1 | tail -2; cat gpurun_out/tex_carry_step_vs_reassembled.txt; done
