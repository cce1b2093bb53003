mkdir -p gpurun_out/r03p
for lib in "" quantum-systems_amd/variants/libqs_amd_ahead1.so; do echo "== lib: ${lib:-default (ahead 2 up to 40 orbitals)}"; QS_AMD_LIB=$lib QS_SWEEP_L=25,28,32,33,36,40,44,48,52,55 python tools/pair4c_sweep.py 2>&1 | grep -v amdgpu | cut -c1-75; done
