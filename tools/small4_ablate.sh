#!/bin/bash
# What bounds the small-basis kernel: variants with parts compiled out (QS_SMALL4_ABLATE), same box.  Build first, here
# or on the box:  for v in 0 1 2 4 8 3 7; do bash tools/build_variant.sh ab$v qs_small4.hip -DQS_SMALL4_ABLATE=$v; done
for v in ${QS_ABLATE_SET:-0 1 2 4 8 3 7}; do
  echo "== ablate $v (1 item loads, 2 stores, 4 MFMAs, 8 LDS reads of A): l dtype us"
  QS_AMD_LIB=quantum-systems_amd/variants/libqs_amd_ab$v.so QS_SWEEP_L="8,16,20,24,32" python tools/small4_sweep.py 2>&1 | grep -v "^l \|amdgpu.ids" | awk '{print $1, $2, $3}' | paste - - - - - - - - - - | cut -c1-200
done
