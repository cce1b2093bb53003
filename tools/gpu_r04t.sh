mkdir -p gpurun_out/r04t
export QS_SWEEP_DTYPES=f64 QS_SWEEP_L=176,190,208,224,240,253 QS_SWEEP_TUNE=gemm_strip=2
for mt in 16 10 8 16; do
  echo "# forced strip, QS_STRIP_MAXT=$mt" >> gpurun_out/r04t/maxt.txt
  QS_STRIP_MAXT=$mt timeout -k 10 600 python tools/size_sweep.py 2>&1 | grep -v "amdgpu.ids\|^l dtype" | cut -c1-130 >> gpurun_out/r04t/maxt.txt
done
cat gpurun_out/r04t/maxt.txt
