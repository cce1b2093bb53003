mkdir -p gpurun_out/r04x
for lib in base strip_aux0 base strip_aux0; do
  if [ $lib = base ]; then unset QS_AMD_LIB; else export QS_AMD_LIB=$PWD/quantum-systems_amd/variants/libqs_amd_$lib.so; fi
  echo "# $lib" >> gpurun_out/r04x/aux.txt
  QS_SWEEP_DTYPES=f64 QS_SWEEP_L=80,97,100,112,130,144,160 QS_SWEEP_TUNE=gemm_strip=2,quad4s=0 timeout -k 10 300 python tools/size_sweep.py 2>&1 | grep -v "amdgpu.ids\|^l dtype" | cut -c1-60 >> gpurun_out/r04x/aux.txt
  QS_SWEEP_DTYPES=c128 QS_SWEEP_L=66,72,80,100,112 QS_SWEEP_TUNE=gemm_strip=2,pair4c=0 timeout -k 10 300 python tools/size_sweep.py 2>&1 | grep -v "amdgpu.ids\|^l dtype" | cut -c1-60 >> gpurun_out/r04x/aux.txt
done
cat gpurun_out/r04x/aux.txt
