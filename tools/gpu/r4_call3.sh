#!/bin/bash
# round 4, GPU call 3: ablations of the RI-fwd pass (debug flags of raster_kernel: 1 = every bin empty, 2 = no stores, 4 = no coverage loop, 8 = no record loads)
set +e
O=gpurun_out/r4c3
mkdir -p $O
R="$GRAFT_REPO_ROOT"
cd "$R"
export PYTHONUNBUFFERED=1
for d in 0 1 2 4 8 3; do
  echo "debug=$d" | tee -a $O/ablation.txt
  VHAP_DEBUG=$d VHAP_HIP_LIB=$R/vhap_amd/lib/libvhap_hip_s80.so python tools/quick_bench_raster.py 2>&1 | grep fused | tee -a $O/ablation.txt
done
