#!/usr/bin/env python
"""Side library with ONE source recompiled under extra compiler flags, linked with the product build's other objects:
    python scripts/build_side_flags.py frontend_pk3.hip fe_maxilp -mllvm -amdgpu-sched-strategy=max-ilp
-> scripts/whatif_libs/lib_<name>.so (timed against the product library in one process, e.g. BASE=lib_fe_maxilp.so scripts/ab_fe_libs.py)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tc-resnet_amd"))
import build as B
B.build()
src, name, extra = sys.argv[1], sys.argv[2], sys.argv[3:]
out = os.path.join(ROOT, "scripts", "whatif_libs")
os.makedirs(out, exist_ok=True)
obj = os.path.join(B.OBJDIR, f"side_{name}.o")
subprocess.check_call([B.HIPCC] + B.FLAGS + extra + ["-c", os.path.join(B.CSRC, src), "-o", obj])
others = [os.path.join(B.OBJDIR, os.path.splitext(s)[0] + ".o") for s in B.SOURCES if s != src]
so = os.path.join(out, f"lib_{name}.so")
subprocess.check_call([B.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so, obj] + others)
print(so)
