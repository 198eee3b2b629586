#!/bin/bash
# round 6, call 32: row order vs 16x16/8x8 tiles in the shading backward at probe sizes (is it the width, the power-of-two stride or the face's size in pixels?)
set +e
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp PYTHONUNBUFFERED=1
O=gpurun_out/r6c32; mkdir -p $O
for shp in 16,512,512 16,520,520 16,504,504 16,640,640 16,768,768 8,1000,1000 8,1024,1024 32,384,384 16,512,768 16,768,512; do for f in 16777216 268435456; do
  timeout 200 python tools/kbench.py --config 2 --shape $shp --only deferred_shade_bwd --debug-flags $f 2>&1 | grep -i "deferred_shade" | sed "s/^/shape=$shp flags=$f /"
done; done | tee $O/kbench_probe.txt
timeout 600 python -m pytest tests/test_deferred_gpu.py -q -x 2>&1 | tail -3
