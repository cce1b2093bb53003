#!/bin/bash
# Round-3 evidence in ONE gpurun call (one box): bench lines with the rocprofv3 kernel stats of the same command, the
# mixed route A/B, the configs[4] preset, and the PMC passes that feed roofline.traffic.  Usage: bash tools/gpu_evidence_r03.sh <tag>
set -o pipefail
TAG=${1:-r03e}
ROOTDIR=$(pwd)
OUT=$ROOTDIR/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
line() {   # name, rocprof?, bench args...
  local name=$1 prof=$2; shift 2
  echo "== $name: bench.py $@"
  timeout -k 10 900 python bench.py "$@" > $OUT/$name.json 2> $OUT/$name.err || { tail -5 $OUT/$name.err; return 1; }
  cut -c1-300 $OUT/$name.json
  [ "$prof" = "1" ] || return 0
  (cd /tmp && timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$name -- \
      python3 $ROOTDIR/bench.py "$@" --no-cpu-baseline --no-probes > $OUT/${name}_under_rocprof.json 2> $OUT/${name}_rocprof.err)
  local st=$(find $OUT/prof_$name -name "*kernel_stats*.csv" | head -1)
  [ -n "$st" ] && python3 tools/condense_profile.py $st $OUT/${name}_kernel_stats.csv "rocprofv3 --kernel-trace --stats -- python3 bench.py $* --no-cpu-baseline --no-probes (same gpurun call as ${name}.json)" && head -4 $OUT/${name}_kernel_stats.csv | cut -c1-200
  rm -rf $OUT/prof_$name
}
line bench_l256_f64 1 --steps 10 --warmup 2
line bench_l256_mixed_native 1 --dtype mixed --steps 5 --warmup 1 --no-cpu-baseline
line bench_l256_mixed_cast 0 --dtype mixed --mixed-route cast --steps 5 --warmup 1 --no-cpu-baseline --no-probes
line bench_l128_mixed_native 1 --dtype mixed --orbitals 128 --steps 30 --warmup 3 --no-cpu-baseline
line bench_l128_mixed_cast 0 --dtype mixed --mixed-route cast --orbitals 128 --steps 30 --warmup 3 --no-cpu-baseline --no-probes
line bench_config4_one_gpu 0 --config 4 --steps 3 --warmup 1 --no-cpu-baseline --no-probes
echo "== PMC passes"
bash tools/gpu_pmc_refresh.sh ${TAG} > $OUT/pmc_all.txt 2>&1; grep "@full" gpurun_out/${TAG}_pmc_*/summary.txt | cut -c1-420
