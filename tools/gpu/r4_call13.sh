#!/bin/bash
# round 4, GPU call 13: FLAME PCA texture model (kernels + NativeStep vs oracle), shard-piece timings
set +e
O=gpurun_out/r4c13
mkdir -p $O
cd "$GRAFT_REPO_ROOT"
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_texpca.py -m gpu -q -x > $O/pytest_texpca.log 2>&1; echo rc=$?; tail -25 $O/pytest_texpca.log | cut -c1-260
python tools/shard_pieces.py 2>&1 | tail -2 | tee $O/shard_pieces.txt
