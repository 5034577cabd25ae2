import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `pytest -m gpu` through gpurun)")


@pytest.fixture(scope="session")
def emu_lib():
    """Host-side emulator build of the kernel sources (tests/emu): test infrastructure only."""
    import tcresnet_amd as T
    so = os.path.join(ROOT, "tests", "emu", "_build", "libtcr_emu.so")
    srcs = [os.path.join(ROOT, "tc-resnet_amd", "csrc", f) for f in os.listdir(os.path.join(ROOT, "tc-resnet_amd", "csrc"))]
    srcs += [os.path.join(ROOT, "tests", "emu", "hip", "hip_runtime.h"), os.path.join(ROOT, "include", "tcresnet_hip.h")]
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(s) for s in srcs):
        clang = "/opt/rocm/lib/llvm/bin/clang++"
        if not os.path.exists(clang):
            pytest.skip("clang++ for the emulator build is not available")
        subprocess.check_call([os.path.join(ROOT, "tests", "emu", "build_emu.sh")])
    return T._lib.load_from(so, "emu")


@pytest.fixture(scope="session")
def hip_lib():
    """The gfx950 product library on a real GPU."""
    import torch
    import tcresnet_amd as T
    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    return T._lib.get()
