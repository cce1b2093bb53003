set -e
OUT=gpurun_out/r02u; mkdir -p $OUT; ROOTDIR=$(pwd); export TMPDIR=/tmp
timeout -k 10 300 python bench.py --orbitals 55 --steps 200 --warmup 20 > $OUT/bench_l55.json 2> $OUT/bench_l55.err || { tail $OUT/bench_l55.err; exit 1; }
cut -c1-700 $OUT/bench_l55.json
cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOTDIR/$OUT/prof -- python3 $ROOTDIR/bench.py --orbitals 55 --steps 200 --warmup 20 --no-cpu-baseline --no-probes > $ROOTDIR/$OUT/bench_l55_under_rocprof.json 2> $ROOTDIR/$OUT/rocprof.err
cd $ROOTDIR
for f in $(find $OUT/prof -name "*kernel_stats*.csv" | head -1); do head -3 $f | cut -c1-200; done
timeout -k 10 300 python -m pytest tests/test_gpu_bench_script.py -x -q -m gpu 2>&1 | tail -2
