#!/bin/bash
# Round-4 final evidence, in parts (a gpurun call is limited to 20 minutes).  Usage: bash tools/gpu_evidence_r04_final.sh <part>
set -o pipefail
PART=${1:-1}
OUT=gpurun_out/r04_final
mkdir -p $OUT
export TMPDIR=/tmp
ROOTDIR=$(pwd)
bench_with_stats() {   # name, bench args...
  local name=$1; shift
  timeout -k 10 400 python bench.py --no-cpu-baseline "$@" > $OUT/bench_$name.json 2> $OUT/bench_$name.err || { echo "bench $name failed"; tail -3 $OUT/bench_$name.err; return 1; }
  (cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOTDIR/$OUT/prof_$name -- python3 $ROOTDIR/bench.py --no-cpu-baseline --no-probes "$@" > $ROOTDIR/$OUT/bench_${name}_under_rocprof.json 2> $ROOTDIR/$OUT/rocprof_$name.err)
  f=$(find $OUT/prof_$name -name "*kernel_stats*.csv" | head -1)
  [ -n "$f" ] && cp $f $OUT/kernel_stats_$name.csv
  rm -rf $OUT/prof_$name
  python3 -c "import json; d=json.load(open('$OUT/bench_$name.json')); print('$name', round(d['value'],2), d['unit'], round(d['ms_per_step'],4), 'ms/step frac', round(d['roofline']['frac'],3), d['roofline']['kernel'][:60])"
}
case $PART in
1)  # tests, smoke, the driver's default line (with the CPU baseline at the full size) and its kernel stats
  timeout -k 10 900 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; rc=$?; tail -3 $OUT/pytest_gpu.log; [ $rc -ne 0 ] && exit $rc
  timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1 || { tail -20 $OUT/smoke.log; exit 1; }; tail -1 $OUT/smoke.log
  ;;
2)
  timeout -k 10 700 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err || { tail -5 $OUT/bench_default.err; exit 1; }
  python3 -c "import json; d=json.load(open('$OUT/bench_default.json')); print('default', d['value'], d['ms_per_step'], d['roofline']['frac'], d['cpu_baseline']['value'], d['cpu_baseline']['sample'][:120])"
  bench_with_stats l256_f64 --steps 10 --warmup 2
  bench_with_stats l256_c128 --dtype c128 --steps 4 --warmup 1
  ;;
3)
  bench_with_stats l55_f64 --orbitals 55 --steps 200 --warmup 20
  bench_with_stats l100_f64 --orbitals 100 --steps 50 --warmup 5
  bench_with_stats l128_f64 --orbitals 128 --steps 30 --warmup 3
  bench_with_stats l130_f64 --orbitals 130 --steps 30 --warmup 3
  bench_with_stats l144_f64 --orbitals 144 --steps 30 --warmup 3
  bench_with_stats l160_f64 --orbitals 160 --steps 20 --warmup 3
  bench_with_stats l253_f64 --orbitals 253 --steps 6 --warmup 2
  bench_with_stats l80_c128 --dtype c128 --orbitals 80 --steps 30 --warmup 3
  bench_with_stats l100_c128 --dtype c128 --orbitals 100 --steps 20 --warmup 3
  bench_with_stats l128_c128 --dtype c128 --orbitals 128 --steps 20 --warmup 3
  bench_with_stats l160_c128 --dtype c128 --orbitals 160 --steps 6 --warmup 2
  ;;
4)  # PMC passes of the strip kernels on the final binary
  bash tools/pmc_pass.sh r04_final_pmc_l144 --orbitals 144 > $OUT/pmc_l144_f64.txt 2>&1; echo "rc=$?"
  bash tools/pmc_pass.sh r04_final_pmc_l253 --orbitals 253 > $OUT/pmc_l253_f64.txt 2>&1; echo "rc=$?"
  bash tools/pmc_pass.sh r04_final_pmc_l100_c128 --orbitals 100 --dtype c128 > $OUT/pmc_l100_c128.txt 2>&1; echo "rc=$?"
  QS_FULL_SIZE=1 timeout -k 10 600 python -m pytest tests/test_gpu_full_size.py -x -q -s > $OUT/full_size.log 2>&1; echo "full size rc=$?"; tail -4 $OUT/full_size.log
  ;;
5)  # dispatch guard, fp64, every size
  QS_GUARD_DTYPES=f64 timeout -k 10 1150 python tools/dispatch_guard.py 2>&1 | grep -v amdgpu.ids > $OUT/dispatch_guard_f64.txt; echo "guard f64 rc=$?"; tail -5 $OUT/dispatch_guard_f64.txt
  ;;
6)  # dispatch guard, complex128, every size up to 160
  QS_GUARD_DTYPES=c128 QS_GUARD_L=$(seq -s, 4 160) timeout -k 10 1150 python tools/dispatch_guard.py 2>&1 | grep -v amdgpu.ids > $OUT/dispatch_guard_c128_4_160.txt; echo "guard c128 rc=$?"; tail -5 $OUT/dispatch_guard_c128_4_160.txt
  ;;
7)  # ... and 161 to 224
  QS_GUARD_DTYPES=c128 QS_GUARD_L=$(seq -s, 161 224) timeout -k 10 1150 python tools/dispatch_guard.py 2>&1 | grep -v amdgpu.ids > $OUT/dispatch_guard_c128_161_224.txt; echo "guard c128 rc=$?"; tail -5 $OUT/dispatch_guard_c128_161_224.txt
  ;;
esac
