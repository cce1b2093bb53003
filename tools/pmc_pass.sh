#!/bin/bash
# PMC passes over the headline bench (each counter group in its own run,
# kernel-trace only, as the MI355X guide prescribes).  Usage: bash tools/pmc_pass.sh <tag> [bench args]
set -o pipefail
TAG=${1:-pmc}; shift
ROOTDIR=$(pwd)
OUT=$ROOTDIR/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
run() {  # name, counters...
  local name=$1; shift
  timeout -k 10 500 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -- \
    python3 $ROOTDIR/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-probes $BENCH_ARGS \
    > $OUT/$name.json 2> $OUT/$name.err || { echo "$name failed"; tail -5 $OUT/$name.err; return 1; }
  echo "$name ok"
}
BENCH_ARGS="$@"
run fetch FETCH_SIZE &&
run write WRITE_SIZE &&
run clk GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 &&
run lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_UNALIGNED_STALL SQ_INST_CYCLES_VMEM
cd $ROOTDIR
python3 tools/pmc_summarise.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
# the raw rocprofv3 directories are large; only the summary travels back
for d in fetch write clk lds; do rm -rf $OUT/$d; done
