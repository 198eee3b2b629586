#!/bin/bash
# round 6, call 25: the face detector (S3FD) -- tests, timing at a 512 x 512 and an 802 x 550 frame
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp PYTHONUNBUFFERED=1
O=gpurun_out/r6c25; mkdir -p $O
timeout 900 python -m pytest tests/test_face_detector.py tests/test_landmarks.py -q 2>&1 | tail -12
timeout 600 python tools/sfd_bench.py --torch 2>&1 | tail -6 | tee $O/r06_sfd_bench.txt
