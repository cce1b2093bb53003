#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
out=gpurun_out/one_rank_link_model_e.txt
: > $out
for G in 8 4 2; do
for bw in 150 75 50; do
  echo "## per-row messages, nothing delivered (no per-message fill), G $G link $bw" >> $out
  ABSENT_PEERS_NO_FILL=1 timeout -k 10 300 python tools/config4_one_rank.py --orbitals 256 --world $G --rank 0 --dtype f64 --steps 3 --samples 1 --link-gbs $bw 2>/dev/null | grep -E "^\{" | cut -c1-420 >> $out
done
done
grep -E "^##|ms_per_step" $out | sed 's/.*"ms_per_step": \([0-9.]*\).*/   ms \1/'
