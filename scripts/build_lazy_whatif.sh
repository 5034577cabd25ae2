#!/bin/bash
# side libraries with bwd_lazy.hip compiled -DTCR_LAZY_WHATIF=<mask> (timing experiments, wrong results; never the product build)
# usage: scripts/build_lazy_whatif.sh 0 1 2 ...   -> scripts/whatif_libs/lib_whatif_<mask>.so  (other objects: the product build's)
cd "$(dirname "$0")/.."
python tc-resnet_amd/build.py > /dev/null
mkdir -p scripts/whatif_libs
OBJS=$(ls tc-resnet_amd/build/*.o | grep -v bwd_lazy.o)
for m in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -x hip -Itc-resnet_amd/csrc -Wall -Wno-unused-function -Wno-unused-variable -Wno-pass-failed \
      -DTCR_LAZY_WHATIF=$m -c tc-resnet_amd/csrc/bwd_lazy.hip -o /tmp/bwd_lazy_whatif_$m.o &&
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o scripts/whatif_libs/lib_whatif_$m.so $OBJS /tmp/bwd_lazy_whatif_$m.o && echo "built $m"
done
