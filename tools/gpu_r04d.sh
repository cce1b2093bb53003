set -o pipefail
mkdir -p gpurun_out/r04d
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "strip or 16_byte_items" > gpurun_out/r04d/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -5 gpurun_out/r04d/tests.log | cut -c1-250
[ $rc -ne 0 ] && exit 0
export QS_SWEEP_DTYPES=f64
export QS_SWEEP_L=97,100,105,112,113,120,127,129,130,136,144,150,153,160,171,176,190,193,200,208,210,224,231,240,253,255
for cfg in "gemm_strip=2" "gemm_strip=1"; do
  echo "# QS_SWEEP_TUNE=$cfg" >> gpurun_out/r04d/sweep.txt
  QS_SWEEP_TUNE=$cfg timeout -k 10 600 python tools/size_sweep.py 2>&1 | grep -v amdgpu.ids >> gpurun_out/r04d/sweep.txt || { echo "sweep $cfg failed"; tail -5 gpurun_out/r04d/sweep.txt; }
done
cat gpurun_out/r04d/sweep.txt | cut -c1-150
