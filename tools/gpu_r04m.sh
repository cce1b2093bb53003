set -o pipefail
mkdir -p gpurun_out/r04m
QS_GUARD_DTYPES=f64 QS_GUARD_L=$(seq -s, 57 132) timeout -k 10 900 python tools/dispatch_guard.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r04m/guard_f64_57_132.txt; echo "guard f64 rc=$?"
QS_GUARD_DTYPES=c128 QS_GUARD_L=$(seq -s, 49 130) timeout -k 10 900 python tools/dispatch_guard.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r04m/guard_c128_49_130.txt; echo "guard c128 rc=$?"
cut -c1-170 gpurun_out/r04m/guard_f64_57_132.txt; cut -c1-170 gpurun_out/r04m/guard_c128_49_130.txt
