#!/bin/bash
# Where does a store instruction of sandwich4b_kernel<14> spend its time?  Memory-pipeline counters at l = 55 (for the next round;
# NOTES.md "the stores of the small-basis kernel").  Each group in its own rocprofv3 run, kernel trace only.
set -o pipefail
ROOTDIR=$(pwd)
OUT=$ROOTDIR/gpurun_out/pmc_vmem_l55
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --list-avail > $OUT/avail.txt 2>&1
WISH="SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES TA_BUSY_avr TA_BUSY_max TA_TA_BUSY_sum TA_BUFFER_WRITE_WAVEFRONTS_sum TA_BUFFER_READ_WAVEFRONTS_sum TA_BUFFER_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TA_TCP_STATE_READ_sum TCP_TOTAL_ACCESSES_sum TCP_TOTAL_WRITE_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCC_NC_WRITE_REQ_sum TCP_TCC_UC_WRITE_REQ_sum TCC_REQ_sum TCC_WRITE_sum TCC_WRITEBACK_sum TCC_EA_WRREQ_sum TCC_EA_WRREQ_64B_sum TCC_EA_WRREQ_STALL_sum TCC_TAG_STALL_sum TCC_BUSY_sum GRBM_GUI_ACTIVE"
HAVE=""
for c in $WISH; do grep -q -w "$c" $OUT/avail.txt && HAVE="$HAVE $c"; done
echo "available of the wish list:$HAVE" | tee $OUT/have.txt
set -- $HAVE
i=0
while [ $# -gt 0 ]; do
  grp=""; n=0
  while [ $# -gt 0 ] && [ $n -lt 4 ]; do grp="$grp $1"; shift; n=$((n+1)); done
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/g$i -- python3 $ROOTDIR/tools/small_l_profile.py 55 > $OUT/g$i.log 2> $OUT/g$i.err || { echo "group $i ($grp) failed"; tail -3 $OUT/g$i.err; continue; }
  echo "group $i ok:$grp"
done
cd $ROOTDIR
python3 - <<'PY'
import csv, glob, collections, os
out = "gpurun_out/pmc_vmem_l55"
tot = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(out + "/g*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "sandwich4b" not in row.get("Kernel_Name", ""):
            continue
        k = row["Counter_Name"]
        tot[k][0] += float(row["Counter_Value"]); tot[k][1] += 1
with open(out + "/summary.txt", "w") as fh:
    for k in sorted(tot):
        v, n = tot[k]
        fh.write(f"{k:40s} per launch {v / max(n, 1):14.4g}   (n = {n})\n")
print(open(out + "/summary.txt").read())
PY
for d in $OUT/g*/; do rm -rf $d; done
