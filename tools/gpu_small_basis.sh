#!/bin/bash
# Evidence for BASELINE.json configs[1] (l = 55): tests of the small-basis kernels, bench line, rocprofv3 kernel
# stats of the same command, PMC passes, sweep over l.  Usage (through gpurun): bash tools/gpu_small_basis.sh <tag>
set -o pipefail
TAG=${1:-r02s}
ROOTDIR=$(pwd)
OUT=$ROOTDIR/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_bench_script.py -x -q -m gpu > $OUT/pytest.log 2>&1 || { tail -20 $OUT/pytest.log; exit 1; }
tail -2 $OUT/pytest.log
timeout -k 10 300 python bench.py --orbitals 55 --steps 200 --warmup 20 > $OUT/bench_l55.json 2> $OUT/bench_l55.err || { tail $OUT/bench_l55.err; exit 1; }
cat $OUT/bench_l55.json | cut -c1-900
cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python3 $ROOTDIR/bench.py --orbitals 55 --steps 200 --warmup 20 --no-cpu-baseline --no-probes > $OUT/bench_l55_under_rocprof.json 2> $OUT/rocprof.err
cd $ROOTDIR
for f in $(find $OUT/prof -name "*kernel_stats*.csv" | head -1); do head -5 $f | cut -c1-200; done
bash tools/pmc_tool.sh ${TAG}_pmc tools/small_l_profile.py 55 > $OUT/pmc.txt 2>&1; grep "sandwich4" $OUT/pmc.txt | cut -c1-600
timeout -k 10 500 python tools/sandwich_check.py > $OUT/sweep.txt 2>&1; grep -c DIFFERS $OUT/sweep.txt; grep "l= 3[3-9]\|l= 4\|l= 5\|l= 6" $OUT/sweep.txt | cut -c1-170
