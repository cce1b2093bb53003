set -o pipefail
mkdir -p gpurun_out/r04j
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_bench_script.py -x -q > gpurun_out/r04j/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -15 gpurun_out/r04j/tests.log | cut -c1-300
QS_GUARD_DTYPES=f64 QS_GUARD_L=64,65,66,72,80,88,96,97,100,104,112,120,127,128,129,130,136,144,150,153,160,171,176,190,192,193,200,208,210,224,231,240,253,255,256 timeout -k 10 900 python tools/dispatch_guard.py > gpurun_out/r04j/guard_f64.txt 2>&1; echo "guard rc=$?"; cat gpurun_out/r04j/guard_f64.txt | cut -c1-200
