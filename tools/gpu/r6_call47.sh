#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_landmarks.py tests/test_face_detector.py -q 2>&1 | tail -3
timeout 300 python tools/fan_bench.py --batch 2 2>&1 | tail -1
timeout 300 python tools/fan_bench.py --batch 2 --no-split 2>&1 | tail -1
