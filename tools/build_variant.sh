#!/bin/bash
# Builds a variant of libdq_hip.so in which ONE OR SEVERAL translation units are compiled with extra flags (A-B experiments on the GPU box):
#   tools/build_variant.sh <name> <file.hip[,file2.hip,...]> <extra flags...>   ->  diffusion-deconvolution-dia-msms-data_amd/build/var_<name>/libdq_hip.so
# Use it with DQ_HIP_LIB=<that path>.  DQ_VARIANT_AS=<file.hip> (single file only): the variant source REPLACES that translation unit.
set -e
PKG=$(dirname "$0")/../diffusion-deconvolution-dia-msms-data_amd
NAME=$1; SRCS=$2; shift 2
mkdir -p $PKG/build/var_$NAME
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form=1"
OBJS=$(ls $PKG/build/*.o)
NEW=""; PIDS=""
for SRC in ${SRCS//,/ }; do
  OUTSRC=${DQ_VARIANT_AS:-$SRC}
  rm -f $PKG/build/var_$NAME/$SRC.o  # (a stale object from an earlier run must not survive a failed compile)
  /opt/rocm/bin/hipcc $FLAGS "$@" -x hip -c $PKG/csrc/$SRC -o $PKG/build/var_$NAME/$SRC.o &
  PIDS="$PIDS $!"
  OBJS=$(echo "$OBJS" | grep -v "/$OUTSRC.o")
  NEW="$NEW $PKG/build/var_$NAME/$SRC.o"
done
for p in $PIDS; do wait $p || { echo "compile failed" >&2; exit 1; }; done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $OBJS $NEW -o $PKG/build/var_$NAME/libdq_hip.so
echo built $PKG/build/var_$NAME/libdq_hip.so
