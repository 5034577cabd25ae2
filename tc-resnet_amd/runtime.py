"""Process-wide defaults for the host-side drop-in layer (which C-ABI library / device the model classes use).

Product code never touches this: the default is the gfx950 library on the current CUDA(HIP) device and it fails
loudly without a GPU.  Tests point it at the emulator build to exercise the host logic on CPU."""
from __future__ import annotations

from typing import Optional

from . import _lib

_default_lib: Optional[_lib.Library] = None
_default_device = None


def set_default(lib: Optional[_lib.Library] = None, device=None) -> None:
    global _default_lib, _default_device
    _default_lib, _default_device = lib, device


def default_lib() -> Optional[_lib.Library]:
    return _default_lib


def default_device():
    return _default_device
