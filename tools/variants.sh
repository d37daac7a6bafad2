#!/bin/bash
# Build variants of libtfk.so that differ in ONE translation unit's compile flags (tuning experiments):
#   bash tools/variants.sh <name> <source.hip> '<extra hipcc flags>'  ->  torchflows_amd/lib/variants/libtfk_<name>.so
# run one with TORCHFLOWS_AMD_LIB=torchflows_amd/lib/variants/libtfk_<name>.so python bench.py ...
set -e
cd "$(dirname "$0")/../torchflows_amd/csrc"
NAME=$1; SRC=$2; EXTRA=$3
mkdir -p ../lib/variants
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -Wno-unused-parameter"
/opt/rocm/bin/hipcc $FLAGS $EXTRA -c -o ../lib/variants/${NAME}_$(basename $SRC .hip).o $SRC
OBJS=$(ls ../lib/obj/*.o | grep -v "/$(basename $SRC .hip).o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/variants/libtfk_$NAME.so $OBJS ../lib/variants/${NAME}_$(basename $SRC .hip).o
ls -la ../lib/variants/libtfk_$NAME.so
