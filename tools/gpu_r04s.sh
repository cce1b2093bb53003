set -o pipefail
mkdir -p gpurun_out/r04s
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "strip" > gpurun_out/r04s/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -15 gpurun_out/r04s/tests.log | cut -c1-300
[ $rc -ne 0 ] && exit 0
QS_GUARD_DTYPES=c128 QS_GUARD_L=129,130,136,144,150,153,160,171,176,182,190,192,200,208,224 timeout -k 10 900 python tools/dispatch_guard.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r04s/guard_c128_129_224.txt; echo "guard rc=$?"; cut -c1-170 gpurun_out/r04s/guard_c128_129_224.txt
QS_GUARD_DTYPES=f64 QS_GUARD_L=256,257,264,272,288,300,320 timeout -k 10 900 python tools/dispatch_guard.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r04s/guard_f64_256_320.txt; echo "guard rc=$?"; cut -c1-170 gpurun_out/r04s/guard_f64_256_320.txt
