#!/bin/bash
# round 4, GPU call 8: host-side timeline of the stage loop (before the feed table), flag-16 row order check
set +e
O=gpurun_out/r4c8
mkdir -p $O
cd "$GRAFT_REPO_ROOT"
export PYTHONUNBUFFERED=1
timeout 300 python tools/stage_timeline.py --frames 256 --epochs 5 --out $O/stage_timeline_before.txt 2>&1 | tail -12
timeout 300 python tools/stage_timeline.py --frames 64 --epochs 3 --out $O/stage_timeline_before_64x3.txt 2>&1 | tail -12
for d in 0 16; do VHAP_DEBUG=$d python tools/quick_bench_raster.py 2>&1 | grep "fused=True" | sed "s/^/debug=$d /"; done | tee $O/rowmul.txt
