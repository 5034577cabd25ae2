"""tc-resnet_amd -- MI355X-native (gfx950) TC-ResNet keyword-spotting hot path.

The directory name carries a hyphen (it is the name the build was given); import it as
`tcresnet_amd` through the shim module at the repository root, or with
`importlib.import_module("tc-resnet_amd")`.
"""
from . import _lib
from ._lib import TcrError
from .engine import DSCNN, Frontend, Graph2D, TCResNet, features_to_planar

__all__ = ["_lib", "TcrError", "Frontend", "TCResNet", "DSCNN", "Graph2D", "features_to_planar"]
