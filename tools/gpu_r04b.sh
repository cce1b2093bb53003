set -o pipefail
mkdir -p gpurun_out/r04b
export TMPDIR=/tmp
./tools/bin/probe_unaligned > gpurun_out/r04b/unaligned.txt 2>&1; echo "probe rc=$?"; cat gpurun_out/r04b/unaligned.txt
timeout -k 10 900 python -m pytest tests/test_gpu_async_transport.py -q -k "missing" > gpurun_out/r04b/async.log 2>&1; echo "async rc=$?"; tail -60 gpurun_out/r04b/async.log | cut -c1-300
