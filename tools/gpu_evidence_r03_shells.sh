#!/bin/bash
set -o pipefail
TAG=r03fin3
ROOTDIR=$(pwd)
OUT=$ROOTDIR/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
line() {
  local name=$1; shift 1
  echo "== $name: bench.py $@"
  timeout -k 10 600 python bench.py "$@" > $OUT/$name.json 2> $OUT/$name.err || { tail -5 $OUT/$name.err; return 1; }
  cut -c1-200 $OUT/$name.json
  (cd /tmp && timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$name -- \
      python3 $ROOTDIR/bench.py "$@" --no-cpu-baseline --no-probes > $OUT/${name}_under_rocprof.json 2> $OUT/${name}_rocprof.err)
  local st=$(find $OUT/prof_$name -name "*kernel_stats*.csv" | head -1)
  [ -n "$st" ] && python3 tools/condense_profile.py $st $OUT/${name}_kernel_stats.csv "rocprofv3 --kernel-trace --stats -- python3 bench.py $* --no-cpu-baseline --no-probes (same gpurun call as ${name}.json)" && head -3 $OUT/${name}_kernel_stats.csv | cut -c1-160
  rm -rf $OUT/prof_$name
}
line bench_l66_f64 --orbitals 66 --steps 100 --warmup 10 --no-cpu-baseline
line bench_l78_f64 --orbitals 78 --steps 100 --warmup 10 --no-cpu-baseline
line bench_l91_f64 --orbitals 91 --steps 60 --warmup 6 --no-cpu-baseline
