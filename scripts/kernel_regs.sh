#!/bin/sh
# Register / LDS / scratch / occupancy line of every kernel in one source file (or of those matching $2):
#   sh scripts/kernel_regs.sh dscnn.hip [name-substring]
# Compiles the file for gfx950 with the library's flags and prints the compiler's resource-usage remarks; no GPU needed.
set -e
D=$(cd "$(dirname "$0")/.." && pwd)
SRC=$D/tc-resnet_amd/csrc/$1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -x hip -I"$D/include" -I"$D/tc-resnet_amd/csrc" \
    -Rpass-analysis=kernel-resource-usage -c "$SRC" -o /dev/null 2>&1 | python3 -c '
import sys, re
pat = sys.argv[1] if len(sys.argv) > 1 else ""
name, row = None, {}
def flush():
    if name and pat in name:
        print("%-80s vgpr %3s agpr %3s scratch %4s lds %6s waves/SIMD %s" % (name[:80], row.get("VGPRs"), row.get("AGPRs"),
              row.get("ScratchSize [bytes/lane]"), row.get("LDS Size [bytes/block]"), row.get("Occupancy [waves/SIMD]")))
for line in sys.stdin:
    m = re.search(r"remark: (?:Function Name: (\S+)|\s*([A-Za-z /\[\]]+): (\d+))", line)
    if not m: continue
    if m.group(1): flush(); name, row = m.group(1), {}
    else: row[m.group(2).strip()] = m.group(3)
flush()
' "$2" | c++filt | cut -c1-200
