#!/bin/bash
# Per-pass kernel durations of the small-basis transform for several builds of the library in ONE gpurun call.
# Usage: bash tools/s4_kernel_ab.sh <l> <variant|base> ...
L=$1; shift
ROOT=$(pwd); export TMPDIR=/tmp
for v in "$@"; do
  if [ $v = base ]; then unset QS_AMD_LIB; else export QS_AMD_LIB=$ROOT/quantum-systems_amd/variants/libqs_amd_$v.so; fi
  rm -rf $ROOT/gpurun_out/s4ab_$v
  (cd /tmp && timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $ROOT/gpurun_out/s4ab_$v -- python3 $ROOT/tools/s4_kernel_time.py $L 200 > $ROOT/gpurun_out/s4ab_$v.log 2>&1)
  grep "per transform" $ROOT/gpurun_out/s4ab_$v.log || tail -5 $ROOT/gpurun_out/s4ab_$v.log
  f=$(find $ROOT/gpurun_out/s4ab_$v -name "*kernel_trace.csv" | head -1)
  echo "== $v"; python3 tools/s4_kernel_time.py --parse $f
done
