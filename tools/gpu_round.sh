#!/bin/bash
# One GPU-box session: parity tests, smoke, bench, rocprof kernel trace.
# Usage (through gpurun): bash tools/gpu_round.sh <tag>
set -o pipefail
TAG=${1:-r01}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== pytest -m gpu" 
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; rc=$?
tail -5 $OUT/pytest_gpu.log
[ $rc -ne 0 ] && exit $rc
echo "== smoke"
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1 || { tail -20 $OUT/smoke.log; exit 1; }
tail -2 $OUT/smoke.log
echo "== bench"
timeout -k 10 600 python bench.py --steps 10 --warmup 2 > $OUT/bench.json 2> $OUT/bench.err || { tail -20 $OUT/bench.err; exit 1; }
cat $OUT/bench.json
echo "== rocprof kernel trace"
ROOTDIR=$(pwd)
cd /tmp && timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOTDIR/$OUT/prof -- python3 $ROOTDIR/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-probes > $ROOTDIR/$OUT/rocprof_bench.json 2> $ROOTDIR/$OUT/rocprof.err; rc=$?
cd $ROOTDIR
echo "rocprof rc=$rc"
find $OUT/prof -name "*kernel_stats*.csv" | head -3
for f in $(find $OUT/prof -name "*kernel_stats*.csv" | head -1); do head -12 $f; done
