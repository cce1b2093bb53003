#!/bin/bash
# Build a variant of libqs_amd.so with extra hipcc flags for ONE source file (kernel A/B inside one gpurun call).
# Usage: bash tools/build_variant.sh <name> <source.hip> <flags...>   -> quantum-systems_amd/variants/libqs_amd_<name>.so
# (run __graft_entry__.build() first: the other objects are taken from csrc/build/)
set -e
NAME=$1; SRC=$2; shift 2
ROOT=$(cd $(dirname $0)/.. && pwd)
CS=$ROOT/quantum-systems_amd/csrc
mkdir -p $ROOT/quantum-systems_amd/variants $CS/build/variants
OBJ=$CS/build/variants/${SRC%.hip}.$NAME.o
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -I$ROOT/include -I$CS "$@" -c $CS/$SRC -o $OBJ 2>/dev/null
OTHERS=""
for f in $CS/build/*.o; do
  [ "$(basename $f)" = "${SRC%.hip}.o" ] || OTHERS="$OTHERS $f"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/quantum-systems_amd/variants/libqs_amd_$NAME.so $OBJ $OTHERS
echo built $NAME
