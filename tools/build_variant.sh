#!/bin/bash
# dev: builds a variant of libgandtr_hip.so into tmpbin/lib_<name>.so: one source file recompiled with extra flags, the other objects as built
#   tools/build_variant.sh <name> <file.hip> [-Dflags ...]
set -e
NAME=$1; SRC=$2; shift 2
cd "$(dirname "$0")/../gandtr_amd/csrc"
mkdir -p ../../tmpbin
OBJ=/tmp/variant_${NAME}_$(basename $SRC .hip).o
EXTRA=""; case "$(basename $SRC)" in conv3x3_halo_c16.hip) EXTRA="-mllvm -pragma-unroll-threshold=400000";; esac      # (as in the Makefile)
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-result $EXTRA "$@" -c $SRC -o $OBJ
OBJS=$(ls *.o | grep -v "^$(basename $SRC .hip).o$")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tmpbin/lib_${NAME}.so $OBJS $OBJ
echo built tmpbin/lib_${NAME}.so
