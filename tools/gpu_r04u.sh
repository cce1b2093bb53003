set -o pipefail
mkdir -p gpurun_out/r04u
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "strip" > gpurun_out/r04u/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -8 gpurun_out/r04u/tests.log | cut -c1-300
[ $rc -ne 0 ] && exit 0
export QS_SWEEP_DTYPES=f64 QS_SWEEP_L=97,100,112,129,130,144,153,160,192,224,253 QS_SWEEP_TUNE=gemm_strip=2
for lib in base strip_nopair base strip_nopair; do
  if [ $lib = base ]; then unset QS_AMD_LIB; else export QS_AMD_LIB=$PWD/quantum-systems_amd/variants/libqs_amd_$lib.so; fi
  echo "# $lib" >> gpurun_out/r04u/pair.txt
  timeout -k 10 600 python tools/size_sweep.py 2>&1 | grep -v "amdgpu.ids\|^l dtype" | cut -c1-60 >> gpurun_out/r04u/pair.txt
done
cat gpurun_out/r04u/pair.txt
