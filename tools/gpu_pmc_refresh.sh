#!/bin/bash
# PMC passes (FETCH_SIZE / WRITE_SIZE / clocks / LDS, one group per run) over the bandwidth workloads and the complex128
# headline kernel, so that their bench lines can carry roofline.traffic.  Usage: bash tools/gpu_pmc_refresh.sh <tag>
set -o pipefail
TAG=${1:-r03}
mkdir -p gpurun_out/$TAG
bash tools/pmc_pass.sh ${TAG}_pmc_antisym --workload antisymmetrize > gpurun_out/$TAG/pmc_antisym.txt 2>&1; echo "antisym rc=$?"
bash tools/pmc_pass.sh ${TAG}_pmc_spin_expand --workload spin_expand > gpurun_out/$TAG/pmc_spin_expand.txt 2>&1; echo "spin_expand rc=$?"
bash tools/pmc_pass.sh ${TAG}_pmc_c128 --dtype c128 > gpurun_out/$TAG/pmc_c128.txt 2>&1; echo "c128 rc=$?"
bash tools/pmc_tool.sh ${TAG}_pmc_spin2 tools/spin2_rate.py > gpurun_out/$TAG/pmc_spin2.txt 2>&1; echo "spin2 rc=$?"
tail -n 30 gpurun_out/$TAG/pmc_*.txt | cut -c1-300
