#!/bin/bash
# round 4, GPU call 7: where the new binning kernel's time goes (ablations: 16384 no pair records, 32768 no head atomics, 16 caller's order, 1024 fat workgroups)
set +e
O=gpurun_out/r4c7
mkdir -p $O
cd "$GRAFT_REPO_ROOT"
export PYTHONUNBUFFERED=1
python tools/raster_probe.py 0 16384 32768 49152 16 1024 1 2 3 2>&1 | grep debug= | tee $O/probe.txt
