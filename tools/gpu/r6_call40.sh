#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp PYTHONUNBUFFERED=1
for i in 1 2 3; do timeout 900 python -m pytest tests/test_deferred_gpu.py -q 2>&1 | grep -E "^E  .*assert|AssertionError|passed|failed" | cut -c1-250 | head -8; done
