#!/bin/bash
# A/B of several builds of libvhap_hip.so on ONE box: per-kernel average durations of the RI-fwd pass under rocprofv3.
#   tools/ab_libs.sh base p4 ...     (vhap_amd/lib/libvhap_hip_<name>.so; see VHAP_HIP_LIB in vhap_amd/_lib.py)
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
for rep in 1 2; do
for v in "$@"; do
  out=/tmp/ab_$v_$rep; rm -rf $out
  VHAP_HIP_LIB=$PWD/vhap_amd/lib/libvhap_hip_$v.so rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python tools/quick_bench_raster.py > /dev/null 2>&1
  f=$(find $out -name "*kernel_stats.csv" | head -1)
  echo -n "$v: "
  python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
out = []
for r in rows:
    n = r["Name"]
    if "raster_kernel" in n or "bin_build" in n:
        tag = "raster<interp>" if "Lb1" in n or "<true>" in n else ("raster" if "raster_kernel" in n else "bin_build")
        out.append(f"{tag} {float(r['AverageNs'])/1e3:.1f} us (x{r['Calls']})")
print("  ".join(sorted(out)))
PY
done
done
