#!/bin/bash
# round 4, GPU call 6: the rebuilt rasteriser (pair records + heads, ordered triangles, kb blocks per wave): bit-exact tests, then timing sweep
set +e
O=gpurun_out/r4c6
mkdir -p $O
R="$GRAFT_REPO_ROOT"
cd "$R"
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_raster_gpu.py -m gpu -q -x > $O/pytest_raster.log 2>&1; echo rc=$?
tail -15 $O/pytest_raster.log
for d in 0 16 1024; do
  echo "debug=$d (256: kb=1, 512: kb=2, 0/768: kb=4; 16: caller's triangle order; 1024: 1024-triangle binning workgroups)" | tee -a $O/sweep.txt
  VHAP_DEBUG=$d timeout 120 python tools/quick_bench_raster.py 2>&1 | grep "fused=" | tee -a $O/sweep.txt
done
cd /tmp && export TMPDIR=/tmp
timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o ri -- python $R/tools/quick_bench_raster.py > /dev/null 2>&1
cd "$R"
cp $O/prof/*kernel_stats.csv $O/ri_kernel_stats.csv 2>/dev/null
python - <<'PY'
import csv, glob
for f in glob.glob('gpurun_out/r4c6/ri_kernel_stats.csv'):
    for r in csv.DictReader(open(f)):
        if 'raster_kernel' in r['Name'] or 'bin_build' in r['Name']:
            print(r['Name'][:70], r['Calls'], float(r['AverageNs'])/1e3, 'us')
PY
echo "== variant nb (no forced 8 waves per SIMD for modes 0/1)"
VHAP_HIP_LIB=$R/vhap_amd/lib/libvhap_hip_nb.so python tools/quick_bench_raster.py 2>&1 | grep "fused=" | tee -a $O/sweep.txt
