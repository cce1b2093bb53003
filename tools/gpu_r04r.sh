set -o pipefail
mkdir -p gpurun_out/r04r
export TMPDIR=/tmp
for l in 55 56 48 64; do
  for lib in base s4b_noloads s4b_nostores s4b_nomem base; do
    if [ $lib = base ]; then unset QS_AMD_LIB; else export QS_AMD_LIB=$PWD/quantum-systems_amd/variants/libqs_amd_$lib.so; fi
    echo "$lib: $(python tools/small_l_profile.py $l 2>&1 | grep -v amdgpu.ids)" >> gpurun_out/r04r/s4b_ablate.txt
  done
done
unset QS_AMD_LIB
cat gpurun_out/r04r/s4b_ablate.txt
bash tools/pmc_pass.sh r04r_pmc_l55 --orbitals 55 > gpurun_out/r04r/pmc_l55.txt 2>&1; echo "pmc l55 rc=$?"
bash tools/pmc_pass.sh r04r_pmc_l56 --orbitals 56 > gpurun_out/r04r/pmc_l56.txt 2>&1; echo "pmc l56 rc=$?"
grep -A2 "== clk\|== fetch\|== write\|== lds" gpurun_out/r04r/pmc_l55.txt | grep -v transpose | cut -c1-600
python bench.py --orbitals 55 --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/r04r/bench_l55.json 2>/dev/null; python -c "import json; d=json.load(open('gpurun_out/r04r/bench_l55.json')); print('bench l=55', d['value'], d['ms_per_step'], d['roofline']['frac'])"
