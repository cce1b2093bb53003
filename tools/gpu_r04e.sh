set -o pipefail
mkdir -p gpurun_out/r04e
bash tools/pmc_pass.sh r04e_pmc_l144 --orbitals 144 > gpurun_out/r04e/pmc_l144.txt 2>&1; echo "rc=$?"
grep -A3 "== clk\|== lds\|== fetch\|== write" gpurun_out/r04e/pmc_l144.txt | grep -v transpose | cut -c1-700
