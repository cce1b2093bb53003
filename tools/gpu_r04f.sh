set -o pipefail
mkdir -p gpurun_out/r04f
export QS_SWEEP_DTYPES=f64 QS_SWEEP_L=112,144,208 QS_SWEEP_TUNE=gemm_strip=2
for lib in base strip_noA strip_noB strip_localst strip_localst_noloads strip_nostores strip_nomem base; do
  if [ $lib = base ]; then unset QS_AMD_LIB; else export QS_AMD_LIB=$PWD/quantum-systems_amd/variants/libqs_amd_$lib.so; fi
  echo "# $lib" >> gpurun_out/r04f/ablate2.txt
  timeout -k 10 300 python tools/size_sweep.py 2>&1 | grep -v "amdgpu.ids\|^l dtype" | cut -c1-60 >> gpurun_out/r04f/ablate2.txt
done
cat gpurun_out/r04f/ablate2.txt
