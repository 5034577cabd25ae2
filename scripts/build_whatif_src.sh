#!/bin/bash
# side libraries with ONE kernel source compiled with a timing what-if macro (wrong results; never the product build)
# usage: scripts/build_whatif_src.sh train_fused TCR_PHASE_WHATIF 0 1 2 ...  -> scripts/whatif_libs/lib_whatif_<mask>.so
cd "$(dirname "$0")/.."
SRC=$1; MACRO=$2; shift 2
EXT=hip; [ -f tc-resnet_amd/csrc/$SRC.cpp ] && EXT=cpp
python tc-resnet_amd/build.py > /dev/null
mkdir -p scripts/whatif_libs
OBJS=$(ls tc-resnet_amd/build/*.o | grep -v "/$SRC.o")
for m in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -x hip -Iinclude -Itc-resnet_amd/csrc -Wall -Wno-unused-function -Wno-unused-variable -Wno-pass-failed \
      -D$MACRO=$m -c tc-resnet_amd/csrc/$SRC.$EXT -o /tmp/${SRC}_whatif_$m.o &&
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o scripts/whatif_libs/lib_whatif_$m.so $OBJS /tmp/${SRC}_whatif_$m.o && echo "built $m"
done
