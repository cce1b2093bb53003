set -o pipefail
mkdir -p gpurun_out/r04l
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "strip or 16_byte_items" > gpurun_out/r04l/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -15 gpurun_out/r04l/tests.log | cut -c1-300
[ $rc -ne 0 ] && exit 0
export QS_SWEEP_DTYPES=c128 QS_SWEEP_L=57,60,64,66,72,78,80,88,91,96,100,105,112,120,127,128
for cfg in "gemm_strip=0" "gemm_strip=2,pair4c=0" "gemm_strip=1"; do
  echo "# QS_SWEEP_TUNE=$cfg" >> gpurun_out/r04l/sweep_c128.txt
  QS_SWEEP_TUNE=$cfg timeout -k 10 600 python tools/size_sweep.py 2>&1 | grep -v "amdgpu.ids\|^l dtype" | cut -c1-130 >> gpurun_out/r04l/sweep_c128.txt
done
cat gpurun_out/r04l/sweep_c128.txt
