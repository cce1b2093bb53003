#!/bin/bash
# the dispatch guard beyond the sizes of profiles/r04_final_dispatch_guard_*: complex128 225-260 (fp64 264-320 ran before)
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
QS_GUARD_L=$(seq -s, 225 260) QS_GUARD_DTYPES=c128 python tools/dispatch_guard.py > gpurun_out/guard_c128_225_260.txt 2> gpurun_out/guard_c128_225_260.err
echo "c128 rc=$?"
tail -n 3 gpurun_out/guard_c128_225_260.txt
