"""CPU: the gfx950 C-ABI library builds, loads and exports every symbol of include/tcresnet_hip.h; the
host-only entry points (no kernel launches) behave as documented."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import tcresnet_amd as T
from oracle import numpy_ref as R

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as g
    g.build()
    return T._lib.get()


def header_functions():
    src = open(os.path.join(ROOT, "include", "tcresnet_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = set(re.findall(r"\b(tcr_[a-z0-9_]+)\s*\(", src))
    names.discard("tcr_padded_len")     # static inline
    return names


def test_exports_every_declared_symbol(lib):
    declared = header_functions()
    assert len(declared) >= 25
    dll = C.CDLL(T._lib.HIP_LIB_PATH)
    for n in sorted(declared):
        assert hasattr(dll, n), f"{n} declared in include/tcresnet_hip.h but not exported"
    assert declared == set(T._lib.ABI_SYMBOLS), declared ^ set(T._lib.ABI_SYMBOLS)
    assert lib.tcr_abi_version() == T._lib.ABI_VERSION == 3
    assert lib.tcr_kernel_name(0) == b"frontend_pk_kernel" and lib.tcr_kernel_name(999) is None


def test_library_is_gfx950_code_object():
    blob = open(T._lib.HIP_LIB_PATH, "rb").read()
    assert b"gfx950" in blob and b"frontend_kernel" in blob


def test_product_path_has_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(T.TcrError):
        T.Frontend()                    # HIP library present, no GPU -> loud failure, never the oracle


def test_frontend_resolve_and_tables(lib):
    for win, hop, nfft, frames in ((480, 160, 512, 98), (640, 320, 1024, 49)):
        cfg = T._lib.FrontendCfg(16000, 16000, win, hop, 0, 0, 64, 40, 80.0, 7600.0, 0)
        assert lib.tcr_frontend_resolve(C.byref(cfg)) == 0
        assert (cfg.nfft, cfg.n_frames) == (nfft, frames)
        n = lib.tcr_frontend_plan_bytes(C.byref(cfg))
        plan = np.zeros(n // 4, np.float32)
        assert lib.tcr_frontend_plan_init(C.byref(cfg), plan.ctypes.data) == 0
        mel = np.zeros((nfft // 2 + 1, 64), np.float32)
        assert lib.tcr_frontend_plan_mel_matrix(C.byref(cfg), plan.ctypes.data, mel.ctypes.data) == 0
        assert np.array_equal(mel, R.linear_to_mel_weight_matrix(64, nfft // 2 + 1, 16000, 80.0, 7600.0, np.float32))
        dct = np.zeros((64, 40), np.float32)
        assert lib.tcr_frontend_plan_dct_matrix(C.byref(cfg), plan.ctypes.data, dct.ctypes.data) == 0
        assert np.abs(dct - R.dct2_matrix(64, 40)).max() < 1e-7
    # error behaviour: unsupported window, bad mel bins, null pointers -> status + message, no crash
    bad = T._lib.FrontendCfg(16000, 16000, 4000, 160, 0, 0, 64, 40, 80.0, 7600.0, 0)
    assert lib.tcr_frontend_resolve(C.byref(bad)) == -1 and b"fft_length" in lib.tcr_last_error()
    bad = T._lib.FrontendCfg(16000, 16000, 480, 160, 0, 0, 40, 40, 80.0, 7600.0, 0)
    assert lib.tcr_frontend_resolve(C.byref(bad)) == -1 and b"num_mel_bins" in lib.tcr_last_error()
    assert lib.tcr_frontend_resolve(None) == -1


@pytest.mark.parametrize("name,width", [("TCResNet8", 1.0), ("TCResNet8", 1.5), ("TCResNet14", 1.0), ("TCResNet14", 1.5)])
def test_net_layout_matches_reference_variables(lib, name, width):
    cfg = T._lib.TCResNetCfg()
    cfg.scope = name.encode()
    ch = R.tcresnet_channels(name, width)
    cfg.in_channels, cfg.t_in, cfg.num_classes, cfg.n_blocks = 40, 98, 12, len(ch) - 1
    for i, c in enumerate(ch):
        cfg.channels[i] = c
    cfg.bn_decay, cfg.bn_eps = 0.997, 0.001
    h = C.c_void_p()
    assert lib.tcr_tcresnet_create(C.byref(cfg), C.byref(h)) == 0
    p, s = R.init_params(R.make_tcresnet(name, width))
    seen, spans = {}, []
    for i in range(lib.tcr_net_num_tensors(h)):
        ti = T._lib.TensorInfo()
        assert lib.tcr_net_tensor_info(h, i, C.byref(ti)) == 0
        seen[ti.name.decode()] = tuple(ti.shape[j] for j in range(ti.rank))
        spans.append((ti.arena, ti.offset, ti.offset + ti.size, ti.kind))
    want = {k: (v.shape[0], 1, v.shape[1], v.shape[2]) for k, v in p.items() if v.ndim == 3}
    want.update({k: v.shape for k, v in p.items() if v.ndim == 1})
    want.update({k: v.shape for k, v in s.items()})
    assert seen == want                                     # TF variable names + shapes
    # arenas: no overlap, weights (L2-decayed) strictly before gamma/beta
    nd = lib.tcr_net_decay_floats(h)
    for arena in (0, 1):
        iv = sorted((a, b) for ar, a, b, _ in spans if ar == arena)
        assert all(iv[i][1] <= iv[i + 1][0] for i in range(len(iv) - 1))
    assert all((b <= nd) == (kind == 0) for ar, a, b, kind in spans if ar == 0)
    assert lib.tcr_net_param_floats(h) >= max(b for ar, a, b, _ in spans if ar == 0)
    assert lib.tcr_net_out_frames(h) == 13 and lib.tcr_net_feat_channels(h) == ch[-1]
    assert 0 < lib.tcr_net_workspace_bytes(h, 4, 0) < lib.tcr_net_workspace_bytes(h, 4, 1) < lib.tcr_net_workspace_bytes(h, 8, 1)
    assert lib.tcr_net_tensor_info(h, 10 ** 6, C.byref(T._lib.TensorInfo())) == -1
    lib.tcr_net_destroy(h)
    # unsupported topologies are refused with a message
    cfg.num_classes = 100
    assert lib.tcr_tcresnet_create(C.byref(cfg), C.byref(h)) == -1 and b"num_classes" in lib.tcr_last_error()


def test_deploy_frontend_plan_and_dscnn_layout(lib):
    """Host-only entry points of the later additions: the deploy-path filterbank (method 2) and the DS-CNN training ABI."""
    cfg = T._lib.FrontendCfg(16000, 16000, 640, 320, 0, 0, 64, 40, 80.0, 7600.0, 2)
    assert lib.tcr_frontend_resolve(C.byref(cfg)) == 0 and cfg.nfft == 1024 and cfg.n_frames == 49
    plan = np.zeros(lib.tcr_frontend_plan_bytes(C.byref(cfg)) // 4, np.float32)
    assert lib.tcr_frontend_plan_init(C.byref(cfg), plan.ctypes.data) == 0
    mel = np.zeros((513, 64), np.float32)
    assert lib.tcr_frontend_plan_mel_matrix(C.byref(cfg), plan.ctypes.data, mel.ctypes.data) == 0
    hz = 8000.0 / 512
    start, end = int(1.5 + 80.0 / hz), int(7600.0 / hz)
    assert np.all(mel[:start] == 0) and np.all(mel[end + 1:] == 0) and np.all(mel[start:end + 1].sum(1) > 0)
    assert np.all((mel > 0).sum(1) <= 2)                         # every bin feeds at most two adjacent channels
    bad = T._lib.FrontendCfg(16000, 16000, 640, 320, 0, 0, 64, 40, 80.0, 7600.0, 3)
    assert lib.tcr_frontend_resolve(C.byref(bad)) == -1 and b"method" in lib.tcr_last_error()
    dcfg = T._lib.DSCNNCfg(49, 10, 12, 276, 5, 10, 4, 2, 1, 2, 2, 0.96, 0.001)
    h = C.c_void_p()
    assert lib.tcr_dscnn_create(C.byref(dcfg), C.byref(h)) == 0
    assert 0 < lib.tcr_dscnn_workspace_bytes(h, 4) < lib.tcr_dscnn_train_workspace_bytes(h, 4) < lib.tcr_dscnn_train_workspace_bytes(h, 8)
    assert lib.tcr_dscnn_forward_train(h, None, None, None, None, 4, 4, 0.0, None, 0, None, None, None, None) == -1
    assert lib.tcr_dscnn_backward(h, None, None, 4, None, 0, None, None) == -1
    lib.tcr_dscnn_destroy(h)
    # input stage: argument checks happen before any launch
    assert lib.tcr_augment_fwd(None, None, None, None, None, None, None, 4, 16000, None, None) == -1
    assert b"tcr_augment_fwd" in lib.tcr_last_error()
