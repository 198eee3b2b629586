#!/bin/bash
# round 6, call 27: every LDS fragment of a K tile read before its MFMA -- tests, the two networks' timings (and with the 128-pixel form: VHAP_DEBUG=8388608)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp PYTHONUNBUFFERED=1
O=gpurun_out/r6c27; mkdir -p $O
timeout 900 python -m pytest tests/test_face_detector.py tests/test_landmarks.py -q 2>&1 | tail -8
timeout 600 python tools/sfd_bench.py --torch 2>&1 | tail -6 | tee $O/r06_sfd_bench.txt
VHAP_DEBUG=8388608 timeout 600 python tools/sfd_bench.py 2>&1 | tail -3 | sed 's/^/[128-pixel workgroups] /' | tee -a $O/r06_sfd_bench.txt
for b in 2 16; do
  timeout 600 python tools/fan_bench.py --batch $b --torch 2>&1 | tail -2 | tee -a $O/r06_fan_bench.txt
  VHAP_DEBUG=8388608 timeout 600 python tools/fan_bench.py --batch $b 2>&1 | tail -1 | sed 's/^/[128-pixel workgroups] /' | tee -a $O/r06_fan_bench.txt
done
