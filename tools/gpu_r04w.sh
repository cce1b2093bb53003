set -o pipefail
mkdir -p gpurun_out/r04w
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "strip or vs_oracle" > gpurun_out/r04w/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -4 gpurun_out/r04w/tests.log | cut -c1-250
for l in 55 56 64 48; do
  for lib in base s4b_aux2 s4b_aux3 s4b_aux17 base; do
    if [ $lib = base ]; then unset QS_AMD_LIB; else export QS_AMD_LIB=$PWD/quantum-systems_amd/variants/libqs_amd_$lib.so; fi
    echo "$lib: $(python tools/small_l_profile.py $l 2>&1 | grep -v amdgpu.ids)" >> gpurun_out/r04w/s4b_aux.txt
  done
done
unset QS_AMD_LIB
cat gpurun_out/r04w/s4b_aux.txt
QS_GUARD_DTYPES=f64 QS_GUARD_L=72,80,88 python tools/dispatch_guard.py 2>&1 | grep -v amdgpu.ids | cut -c1-150
