set -o pipefail
mkdir -p gpurun_out/r04g
export QS_SWEEP_DTYPES=f64 QS_SWEEP_L=112,130,144,160,208,253 QS_SWEEP_TUNE=gemm_strip=2
for st in 0 2 4 8 16 0; do
  export QS_STRIP_STAGGER=$st
  echo "# stagger $st" >> gpurun_out/r04g/stagger.txt
  timeout -k 10 300 python tools/size_sweep.py 2>&1 | grep -v "amdgpu.ids\|^l dtype" | cut -c1-60 >> gpurun_out/r04g/stagger.txt
done
cat gpurun_out/r04g/stagger.txt
