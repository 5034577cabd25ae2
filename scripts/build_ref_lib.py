#!/usr/bin/env python
"""Builds the library of another git revision into tc-resnet_amd/lib/side/libtcr_<name>.so so that two builds can be timed in ONE
process on the same box (box-to-box spread is +-3 %).  usage: python scripts/build_ref_lib.py <git-ref> [name]"""
import os, subprocess, sys, tempfile, shutil
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ref = sys.argv[1]
name = sys.argv[2] if len(sys.argv) > 2 else ref.replace("/", "_")
tmp = tempfile.mkdtemp(prefix="tcr_ref_")
try:
    subprocess.check_call(f"git -C {ROOT} archive {ref} tc-resnet_amd include | tar -x -C {tmp}", shell=True)
    subprocess.check_call([sys.executable, os.path.join(tmp, "tc-resnet_amd", "build.py")])
    dst = os.path.join(ROOT, "tc-resnet_amd", "lib", "side")
    os.makedirs(dst, exist_ok=True)
    shutil.copy(os.path.join(tmp, "tc-resnet_amd", "lib", "libtcresnet_hip.so"), os.path.join(dst, f"libtcr_{name}.so"))
    print(os.path.join(dst, f"libtcr_{name}.so"))
finally:
    shutil.rmtree(tmp, ignore_errors=True)
