#!/usr/bin/env python
"""Side libraries for timing what-ifs of the fused eval network: fused.hip recompiled with -DTCR_FUSED_WHATIF=<mask>, linked with the
product build's other objects into tc-resnet_amd/lib/whatif/libtcr_w<mask>.so (WRONG results; never loaded by the product loader)."""
import os, subprocess, sys, concurrent.futures as cf
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tc-resnet_amd"))
import build as B
B.build()
masks = [int(m) for m in sys.argv[1:]] or [0, 1, 2, 4, 6, 7, 8, 16, 32, 64, 127]
os.makedirs(os.path.join(B.LIBDIR, "whatif"), exist_ok=True)
others = [os.path.join(B.OBJDIR, os.path.splitext(s)[0] + ".o") for s in B.SOURCES if s != "fused.hip"]
def one(m):
    obj = os.path.join(B.OBJDIR, f"fused_w{m}.o")
    subprocess.check_call([B.HIPCC] + B.FLAGS + [f"-DTCR_FUSED_WHATIF={m}", "-c", os.path.join(B.CSRC, "fused.hip"), "-o", obj])
    so = os.path.join(B.LIBDIR, "whatif", f"libtcr_w{m}.so")
    subprocess.check_call([B.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so, obj] + others)
    return so
with cf.ThreadPoolExecutor(max_workers=6) as ex:
    for so in ex.map(one, masks): print(so)
