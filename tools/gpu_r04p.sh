set -o pipefail
mkdir -p gpurun_out/r04p
export TMPDIR=/tmp
export QS_AMD_LIB=$PWD/quantum-systems_amd/variants/libqs_amd_cx_sb8.so
python bench.py --dtype c128 --orbitals 128 --no-cpu-baseline --no-probes --steps 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('sb8 l=128', round(d['value'],2), d['parity'])"
ROOTDIR=$PWD; cd /tmp
timeout -k 10 500 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS --output-format csv -d $ROOTDIR/gpurun_out/r04p/lds -- python3 $ROOTDIR/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-probes --dtype c128 --orbitals 128 > $ROOTDIR/gpurun_out/r04p/lds.json 2> $ROOTDIR/gpurun_out/r04p/lds.err
cd $ROOTDIR
python3 tools/pmc_summarise.py gpurun_out/r04p | grep gemm_fast | cut -c1-400
rm -rf gpurun_out/r04p/lds
