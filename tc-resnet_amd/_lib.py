"""ctypes binding of include/tcresnet_hip.h.

The product path loads ONLY the hipcc-built gfx950 library (`tc-resnet_amd/lib/libtcresnet_hip.so`,
built by `build.py` / `__graft_entry__.build()`) and raises if it is missing: there is no CPU
fallback.  `load_from(path)` exists so that tests can bind the same prototypes onto the host-side
emulator build of the kernel sources (tests/emu), which is test infrastructure only.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

HALO = 4
MAX_BLOCKS = 16
ABI_VERSION = 3            # == TCR_ABI_VERSION of include/tcresnet_hip.h these prototypes were written against
_HERE = os.path.dirname(os.path.abspath(__file__))
HIP_LIB_PATH = os.path.join(_HERE, "lib", "libtcresnet_hip.so")


class TcrError(RuntimeError):
    pass


class FrontendCfg(C.Structure):
    _fields_ = [("sample_rate", C.c_int32), ("n_samples", C.c_int32), ("win", C.c_int32), ("hop", C.c_int32),
                ("nfft", C.c_int32), ("n_frames", C.c_int32), ("n_mel", C.c_int32), ("n_coef", C.c_int32),
                ("lower_hz", C.c_float), ("upper_hz", C.c_float), ("method", C.c_int32)]


class TCResNetCfg(C.Structure):
    _fields_ = [("scope", C.c_char * 32), ("in_channels", C.c_int32), ("t_in", C.c_int32), ("num_classes", C.c_int32),
                ("n_blocks", C.c_int32), ("channels", C.c_int32 * (MAX_BLOCKS + 1)), ("bn_decay", C.c_float),
                ("bn_eps", C.c_float)]


class DSCNNCfg(C.Structure):
    _fields_ = [("h_in", C.c_int32), ("w_in", C.c_int32), ("num_classes", C.c_int32), ("depth", C.c_int32),
                ("n_separable", C.c_int32), ("conv1_kh", C.c_int32), ("conv1_kw", C.c_int32), ("conv1_sh", C.c_int32),
                ("conv1_sw", C.c_int32), ("ds1_sh", C.c_int32), ("ds1_sw", C.c_int32), ("bn_decay", C.c_float),
                ("bn_eps", C.c_float)]


class TensorInfo(C.Structure):
    _fields_ = [("name", C.c_char * 96), ("kind", C.c_int32), ("arena", C.c_int32), ("offset", C.c_int64),
                ("size", C.c_int64), ("shape", C.c_int32 * 4), ("rank", C.c_int32)]


_P = C.c_void_p
_PROTOTYPES = {
    # name: (restype, argtypes)
    "tcr_abi_version": (C.c_int, []),
    "tcr_last_error": (C.c_char_p, []),
    "tcr_kernel_name": (C.c_char_p, [C.c_int]),
    "tcr_tune": (C.c_int, [C.c_int, C.c_int]),
    "tcr_internal_stream": (_P, [C.c_int, _P]),
    "tcr_frontend_resolve": (C.c_int, [C.POINTER(FrontendCfg)]),
    "tcr_frontend_plan_bytes": (C.c_size_t, [C.POINTER(FrontendCfg)]),
    "tcr_frontend_plan_init": (C.c_int, [C.POINTER(FrontendCfg), _P]),
    "tcr_frontend_plan_mel_matrix": (C.c_int, [C.POINTER(FrontendCfg), _P, _P]),
    "tcr_frontend_plan_dct_matrix": (C.c_int, [C.POINTER(FrontendCfg), _P, _P]),
    "tcr_frontend_fwd": (C.c_int, [C.POINTER(FrontendCfg), _P, _P, C.c_int, _P, _P]),
    "tcr_frontend_fwd_rounds": (C.c_int, [C.POINTER(FrontendCfg), _P, _P, C.c_int, _P, C.c_int, _P]),
    "tcr_features_to_planar": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, _P, _P]),
    "tcr_features_from_planar": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, _P, _P]),
    "tcr_tcresnet_create": (C.c_int, [C.POINTER(TCResNetCfg), C.POINTER(_P)]),
    "tcr_net_destroy": (None, [_P]),
    "tcr_net_param_floats": (C.c_int64, [_P]),
    "tcr_net_decay_floats": (C.c_int64, [_P]),
    "tcr_net_stat_floats": (C.c_int64, [_P]),
    "tcr_net_num_tensors": (C.c_int, [_P]),
    "tcr_net_tensor_info": (C.c_int, [_P, C.c_int, C.POINTER(TensorInfo)]),
    "tcr_net_out_frames": (C.c_int, [_P]),
    "tcr_net_feat_channels": (C.c_int, [_P]),
    "tcr_net_workspace_bytes": (C.c_size_t, [_P, C.c_int, C.c_int]),
    "tcr_net_forward_infer": (C.c_int, [_P, _P, _P, _P, C.c_int, _P, C.c_size_t, _P, _P, _P, _P]),
    "tcr_net_forward_train": (C.c_int, [_P, _P, _P, _P, _P, C.c_int, C.c_int, C.c_float, C.c_uint64, C.c_int64,
                                        C.c_float, _P, C.c_size_t, _P, _P, _P, _P]),
    "tcr_net_frozen_floats": (C.c_int64, [_P]),
    "tcr_net_fold_bn": (C.c_int, [_P, _P, _P, _P, _P]),
    "tcr_net_forward_frozen": (C.c_int, [_P, _P, _P, _P, C.c_int, _P, C.c_size_t, _P, _P, _P, _P]),
    "tcr_forward_waveform": (C.c_int, [C.POINTER(FrontendCfg), _P, _P, _P, _P, _P, C.c_int, _P, C.c_int, _P, _P, C.c_size_t, _P, _P, _P, _P]),
    "tcr_net_backward": (C.c_int, [_P, _P, _P, C.c_int, _P, C.c_size_t, _P, _P]),
    "tcr_net_num_stages": (C.c_int, [_P, C.c_int]),
    "tcr_net_stage_sums": (C.c_int, [_P, C.c_int, C.c_int, _P, C.c_int, C.POINTER(_P), C.POINTER(C.c_int64)]),
    "tcr_net_forward_train_stage": (C.c_int, [_P, _P, _P, _P, _P, C.c_int, C.c_int, C.c_float, C.c_uint64, C.c_int64,
                                              C.c_float, _P, C.c_size_t, _P, _P, _P, C.c_int, _P]),
    "tcr_net_backward_stage": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, _P, C.c_size_t, _P, C.c_int, _P]),
    "tcr_net_num_levels": (C.c_int, [_P, C.c_int]),
    "tcr_net_level_sums": (C.c_int, [_P, C.c_int, C.c_int, _P, C.c_int, C.POINTER(_P), C.POINTER(C.c_int64)]),
    "tcr_net_forward_train_level": (C.c_int, [_P, _P, _P, _P, _P, C.c_int, C.c_int, C.c_float, C.c_uint64, C.c_int64,
                                              C.c_float, _P, C.c_size_t, _P, _P, _P, C.c_int, _P]),
    "tcr_net_backward_level": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, _P, C.c_size_t, _P, C.c_int, _P]),
    "tcr_dscnn_create": (C.c_int, [C.POINTER(DSCNNCfg), C.POINTER(_P)]),
    "tcr_dscnn_destroy": (None, [_P]),
    "tcr_dscnn_param_floats": (C.c_int64, [_P]),
    "tcr_dscnn_stat_floats": (C.c_int64, [_P]),
    "tcr_dscnn_num_tensors": (C.c_int, [_P]),
    "tcr_dscnn_tensor_info": (C.c_int, [_P, C.c_int, C.POINTER(TensorInfo)]),
    "tcr_dscnn_workspace_bytes": (C.c_size_t, [_P, C.c_int]),
    "tcr_dscnn_forward_infer": (C.c_int, [_P, _P, _P, _P, C.c_int, _P, C.c_size_t, _P, _P, _P]),
    "tcr_augment_fwd": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, C.c_int, C.c_int, _P, _P]),
    "tcr_dscnn_train_workspace_bytes": (C.c_size_t, [_P, C.c_int]),
    "tcr_dscnn_forward_train": (C.c_int, [_P, _P, _P, _P, _P, C.c_int, C.c_int, C.c_float, _P, C.c_size_t, _P, _P, _P, _P]),
    "tcr_dscnn_backward": (C.c_int, [_P, _P, _P, C.c_int, _P, C.c_size_t, _P, _P]),
    "tcr_dscnn_num_units": (C.c_int, [_P]),
    "tcr_dscnn_unit_output": (C.c_int, [_P, C.c_int, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "tcr_dscnn_materialize_unit": (C.c_int, [_P, C.c_int, C.c_int, _P, C.c_size_t, _P]),
    "tcr_dscnn_num_stages": (C.c_int, [_P]),
    "tcr_dscnn_stage_sums": (C.c_int, [_P, C.c_int, C.c_int, _P, C.c_int, C.POINTER(_P), C.POINTER(C.c_int64)]),
    "tcr_dscnn_forward_train_stage": (C.c_int, [_P, _P, _P, _P, _P, C.c_int, C.c_int, C.c_float, _P, C.c_size_t, _P, _P, _P, C.c_int, _P]),
    "tcr_dscnn_backward_stage": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, _P, C.c_size_t, _P, C.c_int, _P]),
    "tcr_g2d_create": (C.c_int, [C.c_char_p, C.c_int, C.c_int, C.c_int, C.POINTER(_P)]),
    "tcr_g2d_destroy": (None, [_P]),
    "tcr_g2d_conv": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_char_p,
                               C.c_char_p]),
    "tcr_g2d_batch_norm": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_char_p]),
    "tcr_g2d_pool": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "tcr_g2d_add": (C.c_int, [_P, C.c_int, C.c_int, C.c_int]),
    "tcr_g2d_dropout": (C.c_int, [_P, C.c_int, C.c_float]),
    "tcr_g2d_time_filter": (C.c_int, [_P, C.c_int, C.c_char_p]),
    "tcr_g2d_group_sum": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_char_p]),
    "tcr_g2d_node_shape": (C.c_int, [_P, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "tcr_g2d_node_output": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int)]),
    "tcr_g2d_finalize": (C.c_int, [_P, C.c_int]),
    "tcr_g2d_param_floats": (C.c_int64, [_P]),
    "tcr_g2d_decay_floats": (C.c_int64, [_P]),
    "tcr_g2d_stat_floats": (C.c_int64, [_P]),
    "tcr_g2d_num_tensors": (C.c_int, [_P]),
    "tcr_g2d_num_classes": (C.c_int, [_P]),
    "tcr_g2d_tensor_info": (C.c_int, [_P, C.c_int, C.POINTER(TensorInfo)]),
    "tcr_g2d_workspace_bytes": (C.c_size_t, [_P, C.c_int, C.c_int]),
    "tcr_g2d_input_from_features": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, _P, _P]),
    "tcr_g2d_forward_infer": (C.c_int, [_P, _P, _P, _P, C.c_int, _P, C.c_size_t, _P, _P, _P]),
    "tcr_g2d_forward_train": (C.c_int, [_P, _P, _P, _P, _P, C.c_int, C.c_int, C.c_uint64, C.c_int64, C.c_float, _P, C.c_size_t, _P, _P, _P,
                                        _P]),
    "tcr_g2d_backward": (C.c_int, [_P, _P, _P, C.c_int, C.c_uint64, C.c_int64, _P, C.c_size_t, _P, _P]),
    "tcr_g2d_num_stages": (C.c_int, [_P]),
    "tcr_g2d_stage_sums": (C.c_int, [_P, C.c_int, C.c_int, _P, C.c_int, C.POINTER(_P), C.POINTER(C.c_int64)]),
    "tcr_g2d_forward_train_stage": (C.c_int, [_P, _P, _P, _P, _P, C.c_int, C.c_int, C.c_uint64, C.c_int64, C.c_float, _P, C.c_size_t, _P, _P, _P,
                                              C.c_int, _P]),
    "tcr_g2d_backward_stage": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.c_uint64, C.c_int64, _P, C.c_size_t, _P, C.c_int, _P]),
    "tcr_sgd_momentum_step": (C.c_int, [_P, _P, _P, C.c_int64, C.c_int64, C.c_float, C.c_float, C.c_float, C.c_float, _P]),
    "tcr_adam_step": (C.c_int, [_P, _P, _P, _P, C.c_int64, C.c_int64, C.c_float, C.c_float, C.c_float, C.c_float,
                                C.c_int64, C.c_float, C.c_float, _P]),
    "tcr_rmsprop_step": (C.c_int, [_P, _P, _P, _P, C.c_int64, C.c_int64, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float,
                                   C.c_float, _P]),
    "tcr_ema_step": (C.c_int, [_P, _P, C.c_int64, C.c_float, _P]),
    "tcr_l2_loss": (C.c_int, [_P, C.c_int64, C.c_float, _P, _P]),
    "tcr_xent_loss_sum": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_float, _P, _P, _P]),
}

ABI_SYMBOLS = tuple(_PROTOTYPES.keys())


class Library:
    """A loaded C-ABI library with typed prototypes and status checking."""

    def __init__(self, path: str, kind: str, allow_missing: bool = False):
        self.path = path
        self.kind = kind            # "hip" (gfx950 product build) or "emu" (tests only)
        self._dll = C.CDLL(path)
        if not allow_missing:
            # a stale library would otherwise fail with a bare AttributeError on the first new symbol, or silently ignore new knobs
            ver = getattr(self._dll, "tcr_abi_version", None)
            got = None
            if ver is not None:
                ver.restype, ver.argtypes = C.c_int, []
                got = int(ver())
            if got != ABI_VERSION:
                raise TcrError(f"{path} reports TCR_ABI_VERSION {got}, these bindings need {ABI_VERSION}: rebuild the HIP library "
                               "(`python tc-resnet_amd/build.py --force`, or tests/emu/build_emu.sh for the emulator build)")
        for name, (res, args) in _PROTOTYPES.items():
            if allow_missing and not hasattr(self._dll, name):      # (side libraries of OLDER revisions: scripts/build_ref_lib.py)
                continue
            fn = getattr(self._dll, name)       # AttributeError if the symbol is missing
            fn.restype = res
            fn.argtypes = args
            setattr(self, name, fn)

    def check(self, status: int, what: str = "") -> None:
        if status != 0:
            msg = self.tcr_last_error()
            raise TcrError(f"{what or 'tcresnet_hip'} failed (status {status}): {msg.decode() if msg else ''}")


_HIP: Optional[Library] = None


def get() -> Library:
    """The gfx950 product library.  Raises (loudly) when it has not been built."""
    global _HIP
    if _HIP is None:
        if not os.path.exists(HIP_LIB_PATH):
            raise TcrError(
                f"{HIP_LIB_PATH} not found: the HIP extension is not built. Run `python -c 'import __graft_entry__ as g; "
                "g.build()'` (hipcc --offload-arch=gfx950). There is no CPU fallback for this path.")
        _HIP = Library(HIP_LIB_PATH, "hip")
    return _HIP


def load_from(path: str, kind: str = "emu", allow_missing: bool = False) -> Library:
    return Library(path, kind, allow_missing)


def padded_len(t: int) -> int:
    return t + 2 * HALO
