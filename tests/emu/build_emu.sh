#!/bin/sh
# TEST INFRASTRUCTURE: host (clang++) build of the kernel sources against tests/emu/hip/hip_runtime.h
# (one object per source, compiled in parallel, then linked)
set -e
cd "$(dirname "$0")/../.."
OUT=tests/emu/_build
mkdir -p $OUT/obj
SRC="tc-resnet_amd/csrc"
FILES="tcr_common.cpp frontend_plan.cpp frontend.hip frontend_pk.hip frontend_pk3.hip conv.hip mfma.hip bn.hip head.hip optim.hip net.cpp dscnn.hip dscnn_bwd.hip fused.hip train_fused.hip train_fused_bwd.hip bwd_lazy.hip augment.hip net2d_kernels.hip net2d.cpp"
FLAGS="-std=c++17 -O2 -fPIC -x c++ -I tests/emu -Wall -Wno-unused-function -Wno-unused-variable -Wno-unknown-attributes -Wno-unknown-pragmas -Wno-pass-failed"
printf '%s\n' $FILES | xargs -P "$(nproc 2>/dev/null || echo 4)" -I{} /opt/rocm/lib/llvm/bin/clang++ $FLAGS -c $SRC/{} -o $OUT/obj/{}.o
OBJS=""
for f in $FILES; do OBJS="$OBJS $OUT/obj/$f.o"; done
/opt/rocm/lib/llvm/bin/clang++ -shared -fPIC $OBJS -o $OUT/libtcr_emu.so
