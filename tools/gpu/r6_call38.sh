#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 900 python -m pytest "tests/test_deferred_gpu.py::test_shading_backward_over_the_covered_pixel_list" -q 2>&1 | grep -E "^E  |^tests.*Error|passed|failed|^>" | cut -c1-260 | head -30
