#!/bin/bash
# Final round-3 evidence in ONE gpurun call: the default bench line and the small-basis lines whose kernels changed late in the round
# (tail of the balanced kernel, one wave per column group in the complex kernel), each with the rocprofv3 kernel stats of the same command.
# Usage: bash tools/gpu_evidence_r03_final.sh <tag>
set -o pipefail
TAG=${1:-r03fin}
ROOTDIR=$(pwd)
OUT=$ROOTDIR/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
line() {   # name, bench args...
  local name=$1; shift 1
  echo "== $name: bench.py $@"
  timeout -k 10 600 python bench.py "$@" > $OUT/$name.json 2> $OUT/$name.err || { tail -5 $OUT/$name.err; return 1; }
  cut -c1-260 $OUT/$name.json
  (cd /tmp && timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$name -- \
      python3 $ROOTDIR/bench.py "$@" --no-cpu-baseline --no-probes > $OUT/${name}_under_rocprof.json 2> $OUT/${name}_rocprof.err)
  local st=$(find $OUT/prof_$name -name "*kernel_stats*.csv" | head -1)
  [ -n "$st" ] && python3 tools/condense_profile.py $st $OUT/${name}_kernel_stats.csv "rocprofv3 --kernel-trace --stats -- python3 bench.py $* --no-cpu-baseline --no-probes (same gpurun call as ${name}.json)" && head -4 $OUT/${name}_kernel_stats.csv | cut -c1-200
  rm -rf $OUT/prof_$name
}
line bench_l256_f64 --steps 10 --warmup 2
line bench_l55_f64 --orbitals 55 --steps 200 --warmup 20
line bench_l56_f64 --orbitals 56 --steps 200 --warmup 20 --no-cpu-baseline
line bench_l55_c128 --orbitals 55 --dtype c128 --steps 100 --warmup 10 --no-cpu-baseline
line bench_l48_c128 --orbitals 48 --dtype c128 --steps 100 --warmup 10 --no-cpu-baseline
line bench_l36_c128 --orbitals 36 --dtype c128 --steps 200 --warmup 20 --no-cpu-baseline
line bench_l20_c128 --orbitals 20 --dtype c128 --steps 400 --warmup 40 --no-cpu-baseline
line bench_l20_f64 --orbitals 20 --steps 400 --warmup 40 --no-cpu-baseline
line bench_l32_f64 --orbitals 32 --steps 400 --warmup 40 --no-cpu-baseline
line bench_l55_mixed --orbitals 55 --dtype mixed --steps 100 --warmup 10 --no-cpu-baseline
line bench_l128_f64 --orbitals 128 --steps 30 --warmup 3 --no-cpu-baseline
echo "== api overhead"
timeout -k 10 300 python tools/api_overhead.py 2>&1 | grep -v amdgpu.ids | tee $OUT/api_overhead.txt
