mkdir -p gpurun_out/r04y
for lib in base strip_nt base; do
  if [ $lib = base ]; then unset QS_AMD_LIB; else export QS_AMD_LIB=$PWD/quantum-systems_amd/variants/libqs_amd_$lib.so; fi
  echo "# $lib" >> gpurun_out/r04y/aux.txt
  QS_SWEEP_DTYPES=f64 QS_SWEEP_L=129,153,176,190,208,231,253,264,300 QS_SWEEP_TUNE=gemm_strip=2 timeout -k 10 500 python tools/size_sweep.py 2>&1 | grep -v "amdgpu.ids\|^l dtype" | cut -c1-60 >> gpurun_out/r04y/aux.txt
  QS_SWEEP_DTYPES=c128 QS_SWEEP_L=130,153,200 QS_SWEEP_TUNE=gemm_strip=2 timeout -k 10 500 python tools/size_sweep.py 2>&1 | grep -v "amdgpu.ids\|^l dtype" | cut -c1-60 >> gpurun_out/r04y/aux.txt
done
cat gpurun_out/r04y/aux.txt
