#!/bin/bash
# Round-4 opening evidence on the binary the round starts with: GPU suite, the opt-in full-size element-wise check,
# bench line + kernel stats at l = 256, PMC passes of the headline kernel (l = 256) and of configs[1] (l = 55).
set -o pipefail
TAG=${1:-r04a}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
bash tools/gpu_round.sh $TAG || exit 1
echo "== full size parity"
QS_FULL_SIZE=1 timeout -k 10 900 python -m pytest tests/test_gpu_full_size.py -x -q -s > $OUT/full_size.log 2>&1; echo "full size rc=$?"
tail -5 $OUT/full_size.log
echo "== pmc l=256 f64"
bash tools/pmc_pass.sh ${TAG}_pmc_l256_f64 > $OUT/pmc_l256_f64.txt 2>&1; echo "rc=$?"
echo "== pmc l=55 f64"
bash tools/pmc_pass.sh ${TAG}_pmc_l55_f64 --l 55 > $OUT/pmc_l55_f64.txt 2>&1; echo "rc=$?"
tail -n 12 $OUT/pmc_l256_f64.txt $OUT/pmc_l55_f64.txt | cut -c1-400
