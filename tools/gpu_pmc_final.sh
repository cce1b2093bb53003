#!/bin/bash
# PMC passes of the three BASELINE single-GPU kernels on the round's final binary
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04_final
bash tools/pmc_pass.sh r04_final_pmc_l256_f64 > gpurun_out/r04_final/pmc_l256_f64.txt 2>&1; echo "f64 rc=$?"
bash tools/pmc_pass.sh r04_final_pmc_l256_c128 --dtype c128 > gpurun_out/r04_final/pmc_l256_c128.txt 2>&1; echo "c128 rc=$?"
bash tools/pmc_pass.sh r04_final_pmc_l55_f64 --orbitals 55 > gpurun_out/r04_final/pmc_l55_f64.txt 2>&1; echo "l55 rc=$?"
tail -n 12 gpurun_out/r04_final/pmc_l256_f64.txt | cut -c1-400
