#!/bin/sh
# TEST INFRASTRUCTURE: host (clang++) build of the kernel sources against tests/emu/hip/hip_runtime.h
set -e
cd "$(dirname "$0")/../.."
mkdir -p tests/emu/_build
SRC="tc-resnet_amd/csrc"
/opt/rocm/lib/llvm/bin/clang++ -std=c++17 -O1 -g -fPIC -shared -x c++ -I tests/emu -Wall -Wno-unused-function -Wno-unused-variable \
  -Wno-unknown-attributes -Wno-unknown-pragmas -Wno-pass-failed \
  $SRC/tcr_common.cpp $SRC/frontend_plan.cpp $SRC/frontend.hip $SRC/frontend_pk.hip $SRC/conv.hip $SRC/mfma.hip $SRC/bn.hip $SRC/head.hip $SRC/optim.hip $SRC/net.cpp $SRC/dscnn.hip $SRC/dscnn_bwd.hip $SRC/fused.hip $SRC/train_fused.hip $SRC/train_fused_bwd.hip $SRC/augment.hip $SRC/net2d_kernels.hip $SRC/net2d.cpp \
  -o tests/emu/_build/libtcr_emu.so
