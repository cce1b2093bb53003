#!/bin/bash
# A/B of two builds of the library inside ONE gpurun call: alternating runs of bench.py at l (default 256).
# Usage: bash tools/ab_lib.sh <variant name under quantum-systems_amd/variants> [bench args]
V=$1; shift
for i in 1 2 3; do
  for lib in base $V; do
    if [ $lib = base ]; then unset QS_AMD_LIB; else export QS_AMD_LIB=$PWD/quantum-systems_amd/variants/libqs_amd_$lib.so; fi
    python bench.py --no-cpu-baseline --no-probes --steps 10 --warmup 2 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib', round(d['value'],2), d['unit'], round(d['ms_per_step_median'],3), 'ms median', round(d['ms_per_step_min'],3), 'min', d['parity'])"
  done
done
