#!/bin/bash
# One gpurun call = one box: every committed bench line of the round together with the rocprofv3 kernel stats of the
# SAME command.  Usage: bash tools/gpu_evidence.sh <tag>
set -o pipefail
TAG=${1:-r02e}
ROOTDIR=$(pwd)
OUT=$ROOTDIR/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
line() {   # name, bench args...
  local name=$1; shift
  echo "== $name: bench.py $@"
  timeout -k 10 900 python bench.py "$@" > $OUT/$name.json 2> $OUT/$name.err || { tail -5 $OUT/$name.err; return 1; }
  cut -c1-400 $OUT/$name.json
  (cd /tmp && timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$name -- \
      python3 $ROOTDIR/bench.py "$@" --no-cpu-baseline --no-probes > $OUT/${name}_under_rocprof.json 2> $OUT/${name}_rocprof.err)
  local st=$(find $OUT/prof_$name -name "*kernel_stats*.csv" | head -1)
  [ -n "$st" ] && python3 tools/condense_profile.py $st $OUT/${name}_kernel_stats.csv "rocprofv3 --kernel-trace --stats -- python3 bench.py $* --no-cpu-baseline --no-probes (same gpurun call as ${name}.json)" && head -3 $OUT/${name}_kernel_stats.csv | cut -c1-200
  rm -rf $OUT/prof_$name
}
line bench_l256_c128 --dtype c128 --steps 5 --warmup 1 --no-cpu-baseline
line bench_spin_expand_l256 --workload spin_expand --steps 5 --warmup 1
line bench_antisymmetrize_l256 --workload antisymmetrize --steps 20 --warmup 2
line bench_l128_f64 --orbitals 128 --steps 50 --warmup 5 --no-cpu-baseline
echo "== PMC passes of the headline kernel"
bash tools/pmc_pass.sh ${TAG}_pmc > $OUT/pmc_l256.txt 2>&1; grep "gemm_fast" $OUT/pmc_l256.txt | cut -c1-500
