set -o pipefail
mkdir -p gpurun_out/r04o
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "fast or edge" > gpurun_out/r04o/tests.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/r04o/tests.log | cut -c1-250
bash tools/ab_lib.sh cx_pad --dtype c128 --orbitals 256 --steps 4 --warmup 1 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04o/ab_c128_l256.txt
bash tools/ab_lib.sh cx_pad --dtype c128 --orbitals 128 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04o/ab_c128_l128.txt
bash tools/pmc_pass.sh r04o_pmc_c128 --dtype c128 > gpurun_out/r04o/pmc_c128.txt 2>&1; echo "pmc rc=$?"; grep -A3 "== lds\|== clk" gpurun_out/r04o/pmc_c128.txt | grep "gemm_fast" | cut -c1-600
