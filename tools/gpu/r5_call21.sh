#!/bin/bash
# round 5, GPU call 21 (the last): the per-shard reg_diffuse semantics test
set +e
mkdir -p gpurun_out/r5c21
cd "$GRAFT_REPO_ROOT"
timeout 150 python -m pytest tests/test_dist_gpu.py::test_two_rank_reg_diffuse_is_the_per_shard_term -m gpu -q > gpurun_out/r5c21/pytest.log 2>&1; echo rc=$?
tail -12 gpurun_out/r5c21/pytest.log | cut -c1-250
