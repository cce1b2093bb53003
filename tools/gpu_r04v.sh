set -o pipefail
mkdir -p gpurun_out/r04v
export TMPDIR=/tmp
export QS_SWEEP_DTYPES=c128 QS_SWEEP_L=57,60,64,66,70,72,78,80 QS_SWEEP_TUNE=gemm_strip=2,pair4c=0
for w in 0 1 0 1; do
  echo "# complex128 forced strip, QS_STRIP_WIDE=$w" >> gpurun_out/r04v/cx_wide.txt
  QS_STRIP_WIDE=$w timeout -k 10 600 python tools/size_sweep.py 2>&1 | grep -v "amdgpu.ids\|^l dtype" | cut -c1-120 >> gpurun_out/r04v/cx_wide.txt
done
cat gpurun_out/r04v/cx_wide.txt
timeout -k 10 900 python -m pytest tests -x -q -m gpu --deselect tests/test_gpu_full_size.py -k "sharded or rccl or comm or async or strip" > gpurun_out/r04v/tests.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/r04v/tests.log
