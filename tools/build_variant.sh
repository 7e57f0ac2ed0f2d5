#!/bin/bash
# Builds a variant of libdq_hip.so in which ONE translation unit is compiled with extra flags (A-B experiments on the GPU box):
#   tools/build_variant.sh <name> <file.hip> <extra flags...>   ->  diffusion-deconvolution-dia-msms-data_amd/build/var_<name>/libdq_hip.so
# Use it with DQ_HIP_LIB=<that path>.
set -e
PKG=$(dirname "$0")/../diffusion-deconvolution-dia-msms-data_amd
NAME=$1; SRC=$2; shift 2
OUTSRC=${DQ_VARIANT_AS:-$SRC}
mkdir -p $PKG/build/var_$NAME
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form=1"
/opt/rocm/bin/hipcc $FLAGS "$@" -x hip -c $PKG/csrc/$SRC -o $PKG/build/var_$NAME/$SRC.o
OBJS=$(ls $PKG/build/*.o | grep -v "/$OUTSRC.o")   # (DQ_VARIANT_AS=<file.hip>: the variant source REPLACES that translation unit)
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $OBJS $PKG/build/var_$NAME/$SRC.o -o $PKG/build/var_$NAME/libdq_hip.so
echo built $PKG/build/var_$NAME/libdq_hip.so
