set -o pipefail
mkdir -p gpurun_out/r04q
bash tools/ab_lib.sh cx_sb8 --dtype c128 --orbitals 128 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04q/ab_sb8_l128.txt
bash tools/ab_lib.sh cx_sb8 --dtype c128 --orbitals 256 --steps 4 --warmup 1 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04q/ab_sb8_l256.txt
