#!/bin/bash
# A/B of several builds of libvhap_hip.so on ONE box: un-contended per-kernel durations of the fit step (eager, serial launches).
#   tools/ab_step_libs.sh <kernel-name-regex> base other ...   (vhap_amd/lib/libvhap_hip_<name>.so)
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
pat=$1; shift
for v in "$@"; do
  out=/tmp/abs_$v; rm -rf $out
  VHAP_HIP_LIB=$PWD/vhap_amd/lib/libvhap_hip_$v.so rocprofv3 --kernel-trace --output-format csv -d $out -- python tools/ab_kernels.py 0 > /dev/null 2>&1
  f=$(find $out -name "*kernel_trace.csv" | head -1)
  echo -n "$v: "; python tools/ab_kernels.py --report $f 0 | tr " " "\n" | grep -E "$pat" | tr "\n" " "; echo
done
