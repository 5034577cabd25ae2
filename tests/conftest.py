import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


@pytest.hookimpl(tryfirst=True)
def pytest_cmdline_main(config):
    """The CPU suite (`-m "not gpu"`) spends its time in the wave-64 emulator, one test at a time: ~10 min serially, ~1.5 min over
    the host's cores (round 6; the fiber switch is a dozen instructions, tests/emu/hip/hip_runtime.h).  When pytest-xdist is importable and the caller chose no `-n` itself, run that suite on min(8, cores) workers
    (`-n 0` or TCR_TEST_SERIAL=1 keeps it serial).  GPU runs (`-m gpu`) stay in one process: one device, one set of streams."""
    if hasattr(config, "workerinput") or os.environ.get("TCR_TEST_SERIAL") == "1":
        return None
    opt = config.option
    if getattr(opt, "markexpr", "") .replace(" ", "") != "notgpu" or not hasattr(opt, "numprocesses") or opt.numprocesses is not None:
        return None
    if getattr(opt, "collectonly", False) or getattr(opt, "usepdb", False):
        return None
    workers = min(8, os.cpu_count() or 1)
    if workers > 1:
        opt.numprocesses = workers
    return None


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `pytest -m gpu` through gpurun)")


@pytest.fixture(scope="session")
def emu_lib():
    """Host-side emulator build of the kernel sources (tests/emu): test infrastructure only."""
    import tcresnet_amd as T
    so = os.path.join(ROOT, "tests", "emu", "_build", "libtcr_emu.so")
    srcs = [os.path.join(ROOT, "tc-resnet_amd", "csrc", f) for f in os.listdir(os.path.join(ROOT, "tc-resnet_amd", "csrc"))]
    srcs += [os.path.join(ROOT, "tests", "emu", "hip", "hip_runtime.h"), os.path.join(ROOT, "include", "tcresnet_hip.h")]
    import fcntl
    os.makedirs(os.path.dirname(so), exist_ok=True)
    with open(os.path.join(os.path.dirname(so), ".lock"), "w") as lock:      # xdist workers: one of them builds, the others wait
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(s) for s in srcs):
                clang = "/opt/rocm/lib/llvm/bin/clang++"
                if not os.path.exists(clang):
                    pytest.skip("clang++ for the emulator build is not available")
                subprocess.check_call([os.path.join(ROOT, "tests", "emu", "build_emu.sh")])
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return T._lib.load_from(so, "emu")


@pytest.fixture(scope="session")
def hip_lib():
    """The gfx950 product library on a real GPU."""
    import torch
    import tcresnet_amd as T
    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    return T._lib.get()
