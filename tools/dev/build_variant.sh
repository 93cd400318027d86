#!/bin/bash
# build_variant.sh <name> <source.hip> <-Dflags...> : link a variant libtdgp with one source recompiled with extra defines
set -e
cd "$(dirname "$0")/../.."
NAME=$1; SRC=$2; shift 2
C=3dgp_amd/csrc
O=tools/dev/variants/$NAME.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -fvisibility=hidden -Wno-unused-result "$@" -c $C/$SRC.hip -o $O
OBJS=""
for s in core bias_act upfirdn2d modconv conv_grad camera_rays field sampling render_grad render_fused; do
  if [ "$s" == "$SRC" ]; then OBJS="$OBJS $O"; else OBJS="$OBJS $C/build/$s.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/dev/variants/$NAME.so $OBJS
echo built tools/dev/variants/$NAME.so
