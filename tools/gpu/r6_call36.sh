#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 900 python -m pytest "tests/test_deferred_gpu.py::test_shading_backward_over_the_covered_pixel_list" tests/test_tex_carry_gpu.py::test_captured_step_with_carried_texture_matches_the_reassembled_one -q 2>&1 | grep -v "^  warn\|Warning" | grep -E "^E |Error|assert|^tests|passed|failed" | cut -c1-300 | head -60
