set -o pipefail
mkdir -p gpurun_out/r04i
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "strip or 16_byte_items" > gpurun_out/r04i/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -5 gpurun_out/r04i/tests.log | cut -c1-250
[ $rc -ne 0 ] && exit 0
export QS_SWEEP_DTYPES=f64 QS_SWEEP_L=97,100,112,127,129,130,136,144,150,160,208,253 QS_SWEEP_TUNE=gemm_strip=2
for w in 0 1 0; do
  echo "# forced strip, QS_STRIP_WIDE=$w" >> gpurun_out/r04i/sweep.txt
  QS_STRIP_WIDE=$w timeout -k 10 600 python tools/size_sweep.py 2>&1 | grep -v "amdgpu.ids\|^l dtype" | cut -c1-120 >> gpurun_out/r04i/sweep.txt
done
cat gpurun_out/r04i/sweep.txt
timeout -k 10 900 python -m pytest tests/test_gpu_bench_script.py -x -q > gpurun_out/r04i/bench_tests.log 2>&1; echo "bench tests rc=$?"; tail -15 gpurun_out/r04i/bench_tests.log | cut -c1-300
