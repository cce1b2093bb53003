set -o pipefail
mkdir -p gpurun_out/r04n
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_api.py -x -q -k "plan" > gpurun_out/r04n/plan.log 2>&1; echo "plan rc=$?"; tail -25 gpurun_out/r04n/plan.log | cut -c1-250
for a in "20 random" "12 random" "32 random" "48 random"; do timeout -k 10 200 python tools/api_overhead.py $a 2>&1 | grep -v amdgpu.ids >> gpurun_out/r04n/api_overhead.txt; done; cat gpurun_out/r04n/api_overhead.txt
timeout -k 10 600 python tools/fock_time.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r04n/fock_time.txt; cat gpurun_out/r04n/fock_time.txt
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_products_property.py -x -q -k "vs_oracle or random_transforms" > gpurun_out/r04n/oracle.log 2>&1; echo "oracle rc=$?"; tail -5 gpurun_out/r04n/oracle.log | cut -c1-250
