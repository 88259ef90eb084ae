#!/bin/bash
# Build a variant of the library with extra compiler flags for k_encode.hip only (experiments: register budget, helper count, LDS layout).
#   tools/build_variant.sh NAME [flags...]   ->  build/variants/NAME/libhomer_gpu.so   (build/ is git-ignored but travels to the GPU box)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; shift
CS=$ROOT/homerhevc_amd/csrc
OBJ=/tmp/homer_variant_objs
mkdir -p $OBJ $ROOT/build/variants/$NAME
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -x hip"
SRCS="tables.cpp context.cpp dropin.cpp cmdlist.cpp k_pixel.hip k_transform.hip k_intra.hip k_interp.hip k_loop.hip k_motion.hip k_tuchain.hip k_intrasearch.hip k_tree.hip k_chromasearch.hip k_saooffsets.hip k_subpel.hip k_probe.hip k_primtest.hip"
for s in $SRCS; do
  o=$OBJ/${s%.*}.o
  if [ ! -f $o ] || [ $CS/$s -nt $o ] || [ $CS/common.h -nt $o ] || [ $CS/tables_layout.h -nt $o ] || [ -n "$(find $CS/enc -name "*.h" -newer $o | head -1)" ]; then echo $s; ( /opt/rocm/bin/hipcc $FL -c $CS/$s -o $o ) & fi
  while [ $(jobs -r | wc -l) -ge 6 ]; do sleep 0.5; done
done
wait
/opt/rocm/bin/hipcc $FL "$@" -c $CS/k_encode.hip -o $OBJ/k_encode_$NAME.o
OBJS=""; for s in $SRCS; do OBJS="$OBJS $OBJ/${s%.*}.o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared $OBJS $OBJ/k_encode_$NAME.o -o $ROOT/build/variants/$NAME/libhomer_gpu.so
echo built build/variants/$NAME/libhomer_gpu.so
