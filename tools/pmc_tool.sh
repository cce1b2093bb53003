#!/bin/bash
# PMC passes over an arbitrary tool script (each counter group in its own run, kernel-trace only).
# Usage: bash tools/pmc_tool.sh <tag> tools/<script>.py [args]
set -o pipefail
TAG=$1; shift
ROOTDIR=$(pwd)
OUT=$ROOTDIR/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
SCRIPT=$ROOTDIR/$1; shift
cd /tmp
run() {
  local name=$1; shift
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -- \
    python3 $SCRIPT $TOOL_ARGS > $OUT/$name.log 2> $OUT/$name.err || { echo "$name failed"; tail -5 $OUT/$name.err; return 1; }
  echo "$name ok"
}
TOOL_ARGS="$@"
run fetch FETCH_SIZE &&
run write WRITE_SIZE &&
run clk GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 &&
run occ SQ_WAVES SQ_LEVEL_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAIT_INST_LDS
# memory-pipeline view (optional: a counter name this rocprofv3 does not know only loses this group)
run tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr TA_TA_BUSY_sum || true
run vmem SQ_INST_CYCLES_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_WAIT_INST_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA || true
cd $ROOTDIR
python3 tools/pmc_summarise.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
cp $OUT/tcp/*/*counter_collection.csv $OUT/tcp_counters.csv 2>/dev/null; cp $OUT/vmem/*/*counter_collection.csv $OUT/vmem_counters.csv 2>/dev/null
for d in fetch write clk occ tcp vmem; do rm -rf $OUT/$d; done
