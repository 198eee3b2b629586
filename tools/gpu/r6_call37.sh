#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 900 python -m pytest "tests/test_deferred_gpu.py::test_shading_backward_over_the_covered_pixel_list" -q 2>&1 | tail -3
for i in 1 2 3 4 5 6; do
  timeout 900 python -m pytest tests/test_tex_carry_gpu.py::test_captured_step_with_carried_texture_matches_the_reassembled_one -q 2>&1 | grep -E "^E  .*assert|AssertionError: |passed|failed" | cut -c1-300 | head -6
done
for i in 1 2 3; do
  VHAP_SHADE_LIST=0 timeout 900 python -m pytest tests/test_tex_carry_gpu.py::test_captured_step_with_carried_texture_matches_the_reassembled_one -q 2>&1 | grep -E "^E  .*assert|AssertionError: |passed|failed" | cut -c1-300 | head -6
done
