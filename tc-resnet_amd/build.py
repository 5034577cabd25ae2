"""Builds the gfx950 shared library (C ABI of include/tcresnet_hip.h) in-tree with hipcc.

    python tc-resnet_amd/build.py [--force] [--verbose]

Output: tc-resnet_amd/lib/libtcresnet_hip.so  (git-ignored; travels with the gpurun snapshot).
hipcc cross-compiles for gfx950 without a GPU, so this also is the "does it build" check.
"""
from __future__ import annotations

import concurrent.futures as cf
import hashlib
import os
import subprocess
import sys
from typing import List

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
OBJDIR = os.path.join(HERE, "build")
LIB = os.path.join(LIBDIR, "libtcresnet_hip.so")
SOURCES = ["tcr_common.cpp", "frontend_plan.cpp", "frontend.hip", "frontend_pk.hip", "frontend_pk3.hip", "conv.hip", "mfma.hip", "bn.hip", "head.hip",
           "optim.hip", "net.cpp", "dscnn.hip", "dscnn_bwd.hip", "fused.hip", "train_fused.hip", "train_fused_bwd.hip", "bwd_lazy.hip", "augment.hip", "net2d_kernels.hip", "net2d.cpp"]
HEADERS = ["tcr_common.h", "gfx950_isa.h", "frontend_plan.h", "frontend_args.h", "kernels.h", "net2d.h", os.path.join("..", "..", "include", "tcresnet_hip.h")]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-x", "hip", "-I", CSRC, "-Wall", "-Wno-unused-function",
         "-Wno-unused-variable", "-Wno-pass-failed"] + os.environ.get("TCR_BUILD_EXTRA", "").split()      # (diagnostic builds: extra -D flags)


def _digest() -> str:
    h = hashlib.sha256()
    for f in SOURCES + HEADERS:
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(fh.read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def _compile(src: str, verbose: bool) -> str:
    obj = os.path.join(OBJDIR, os.path.splitext(src)[0] + ".o")
    cmd = [HIPCC] + FLAGS + ["-c", os.path.join(CSRC, src), "-o", obj]
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed on {src}:\n{r.stdout}\n{r.stderr}")
    if verbose and r.stderr.strip():
        print(r.stderr, file=sys.stderr)
    return obj


def build(force: bool = False, verbose: bool = False) -> str:
    os.makedirs(LIBDIR, exist_ok=True)
    os.makedirs(OBJDIR, exist_ok=True)
    stamp = os.path.join(LIBDIR, "libtcresnet_hip.digest")
    dig = _digest()
    if not force and os.path.exists(LIB) and os.path.exists(stamp) and open(stamp).read().strip() == dig:
        return LIB
    with cf.ThreadPoolExecutor(max_workers=min(8, len(SOURCES))) as ex:
        objs: List[str] = list(ex.map(lambda s: _compile(s, verbose), SOURCES))
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    with open(stamp, "w") as fh:
        fh.write(dig)
    return LIB


if __name__ == "__main__":
    try:
        print(build(force="--force" in sys.argv, verbose="--verbose" in sys.argv))
    except Exception as e:      # noqa: BLE001
        print("BUILD FAILED:", str(e)[-1500:])
        sys.exit(1)
