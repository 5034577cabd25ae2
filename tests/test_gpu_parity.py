"""GPU (-m gpu): the gfx950 library, called through the C ABI, against the golden vectors / the oracle, plus
size-independent properties at BASELINE.json's full batch (4096)."""
import numpy as np
import pytest
import torch

import tcresnet_amd as T
from oracle import numpy_ref as R
from tests import common as Cm

pytestmark = pytest.mark.gpu


def test_hip_library_is_the_one_loaded(hip_lib):
    assert hip_lib.kind == "hip" and hip_lib.path.endswith("tc-resnet_amd/lib/libtcresnet_hip.so")
    maps = open("/proc/self/maps").read()
    assert "libtcresnet_hip.so" in maps


@pytest.mark.parametrize("tag", ["3010", "4020"])
def test_frontend(hip_lib, tag):
    Cm.check_frontend(hip_lib, tag)


@pytest.mark.parametrize("tag", ["3010", "4020"])
def test_frontend_edge_rows(hip_lib, tag):
    Cm.check_frontend_edges(hip_lib, tag)


def test_augment_frontend_net_chain_matches_oracle(hip_lib):
    """The composed path the reference feeds its model with (datasets/audio_data_wrapper.py:37-58 -> preprocessors.py:183-194 ->
    tc_resnet.py:57-70): tcr_augment_fwd -> front-end -> TCResNet8 eval forward against augment_ref -> R.mfcc -> R.forward.  The batch
    holds ordinary clips, silent clips (empty file) under background noise at volumes 0.01 / 0.05 / 0.1 and an un-mixed silent clip:
    logits within 1e-4, identical argmax."""
    from oracle import augment_ref as A
    from tcresnet_amd.datasets import augmentation_factory as F
    from tcresnet_amd import runtime
    rng = np.random.RandomState(21)
    desired = 16000
    t = np.arange(20000) / 16000.0
    clips = [np.clip((0.3 * np.sin(2 * np.pi * (300 + 170 * i) * t) + 0.2 * rng.uniform(-1, 1, t.size)) * 32768, -32768, 32767).astype(np.int16)[:n]
             for i, n in enumerate((16000, 12000, 20000, 15999, 16000, 9000))]
    clips += [np.zeros(0, np.int16)]                        # the `_silence_` sample: '' filename -> empty clip -> zeros
    bgs = [rng.randint(-32768, 32768, 3 * desired).astype(np.int16), (rng.randint(-2000, 2000, 2 * desired)).astype(np.int16)]
    idx = np.array([0, 1, 2, 3, 4, 5, 6, 6, 6, 6, 6, 0], np.int64)
    shift = np.array([0, 100, -1600, 1599, -3, 7, 0, 55, -200, 0, 0, 0], np.int32)
    bg_idx = np.array([0, 1, 0, 1, 0, 1, 0, 0, 1, 1, 0, 1], np.int64)
    bg_crop = np.array([0, 5, 1000, 31, 17, 16000, 100, 32000, 777, 3, 0, 0], np.int64)
    vol = np.array([0.0, 0.02, 0.1, 0.07, 0.0, 0.05, 0.01, 0.05, 0.1, 0.1, 0.0, 0.1], np.float32)
    runtime.set_default(hip_lib, "cuda")
    try:
        pool, bg = F.PcmPool(clips), F.PcmPool(bgs)
        fn = F.get_audio_augmentation_fn("anchored_slice_or_pad_with_shift")
        wav = fn(pool, idx, desired, "wav", 16000, background_data=bg, is_training=True, draws=(shift, bg_idx, bg_crop, vol))[..., 0].contiguous()
    finally:
        runtime.set_default(None, None)
    ref_wav = A.augment_batch(pool.data.cpu().numpy(), pool.offsets[idx], pool.lengths[idx], shift, bg.data.cpu().numpy(),
                              bg.offsets[bg_idx] + bg_crop, vol, desired)
    assert np.array_equal(wav.cpu().numpy(), ref_wav)       # (the input stage is bit exact)
    assert np.abs(ref_wav[10]).max() == 0 and 0 < np.abs(ref_wav[6]).max() < 0.011
    for tag, cfg in (("4020", R.FRONTEND_4020), ("3010", R.FRONTEND_3010)):
        arch = R.make_tcresnet("TCResNet8", 1.0)
        p, s = R.init_params(arch, 5)
        R.randomize_bn(arch, p, s, 6)
        fe = Cm.make_frontend(hip_lib, cfg.win, cfg.hop)
        net = Cm.make_net(hip_lib, "TCResNet8", 1.0, fe.n_frames, p, s)
        feat = fe(wav)
        logits, probs = net.forward_infer(feat)
        x = R.mfcc(ref_wav.astype(np.float64), cfg)
        ref = R.forward(arch, p, s, x, False)
        assert np.abs(fe.reference_view(feat)[..., 0].cpu().numpy() - x).max() < Cm.MFCC_TOL, tag
        err = np.abs(logits.cpu().numpy() - ref["logits"]).max()
        assert err < Cm.LOGIT_TOL, f"{tag}: chained logits differ by {err}"
        assert np.array_equal(logits.cpu().numpy().argmax(1), ref["logits"].argmax(1))
        assert np.abs(probs.cpu().numpy() - ref["probs"]).max() < 1e-5


@pytest.mark.parametrize("tag", ["3010", "4020"])
def test_frontend_deploy_path(hip_lib, tag):
    Cm.check_frontend_deploy(hip_lib, tag)
    try:                                    # the general (scalar-FP32) kernel implements the same method
        hip_lib.tcr_tune(1, 4)
        Cm.check_frontend_deploy(hip_lib, tag)
    finally:
        hip_lib.tcr_tune(1, 0)


@pytest.mark.parametrize("win,hop", [(640, 320), (480, 160)])
def test_frontend_persistent_chunks_are_position_independent(hip_lib, win, hop):
    """The packed kernel's persistent workgroups walk chunks of 32-64 frames; an utterance's features must not depend on which
    chunk / workgroup / round its frames land in: rows of a ragged large batch (more chunks than workgroups, last chunk partial) are
    BITWISE the rows of small batches of the same waveforms, and the general scalar kernel agrees within the tolerance."""
    b = 4099
    wav = torch.from_numpy(R.synth_waveforms(64, seed=11)).to("cuda").repeat(65, 1)[:b].contiguous()
    fe = Cm.make_frontend(hip_lib, win, hop)
    big = fe(wav)
    for lo, n in ((0, 7), (2048, 3), (b - 5, 5)):
        assert torch.equal(big[lo:lo + n], fe(wav[lo:lo + n].contiguous())), (lo, n)
    try:
        hip_lib.tcr_tune(1, 4)
        ref = fe(wav[:64].contiguous())
    finally:
        hip_lib.tcr_tune(1, 0)
    assert float((big[:64] - ref).abs().max()) < Cm.MFCC_TOL


def test_frontend_kernels_agree_over_batch_sizes(hip_lib):
    """Chunk / wave / round geometry of the three-waves front-end over awkward batch sizes (one round per chunk at the small ones,
    partial last chunks, fewer chunks than workgroups): bitwise the two-waves kernel, both reference framings."""
    wav_all = torch.from_numpy(R.synth_waveforms(64, seed=3)).to("cuda").repeat(16, 1)
    for win, hop in ((640, 320), (480, 160)):
        fe = Cm.make_frontend(hip_lib, win, hop)
        for b in (1, 2, 3, 7, 8, 9, 15, 16, 17, 63, 64, 65, 127, 129, 1000):
            wav = wav_all[:b].contiguous()
            new = fe(wav).clone()
            try:
                hip_lib.tcr_tune(23, 1)
                old = fe(wav).clone()
            finally:
                hip_lib.tcr_tune(23, 0)
            assert torch.equal(new, old), (win, hop, b)


def test_small_batch_network_kernel_is_bitwise_the_throughput_kernel(hip_lib):
    Cm.check_small_batch_kernel(hip_lib)


@pytest.mark.parametrize("tag", ["3010", "4020"])
def test_edge_rows_logits(hip_lib, tag):
    Cm.check_edge_rows_logits(hip_lib, tag)


@pytest.mark.parametrize("batch", [5, 4099])
def test_frontend_three_wave_kernel_is_bitwise_the_two_wave_kernel(hip_lib, batch):
    Cm.check_frontend_kernels_bitwise(hip_lib, batch)


def test_frontend_variants(hip_lib):
    fx = Cm.load("frontend_4020.npz")
    wav = Cm.to_dev(hip_lib, fx["wav"])
    f10 = Cm.make_frontend(hip_lib, 640, 320, num_mfccs=10)
    got = f10.reference_view(f10(wav))[..., 0].cpu().numpy()
    assert np.abs(got - fx["mfcc"][..., :10]).max() < Cm.MFCC_TOL
    flm = Cm.make_frontend(hip_lib, 640, 320, method="log_mel_spectrogram")
    got = flm.reference_view(flm(wav))[..., 0].cpu().numpy()
    assert np.abs(got[:2] - fx["log_mel_magnitude"][:2]).max() < 1e-4
    x = torch.randn(3, 49, 40, device="cuda")
    planar = T.features_to_planar(x)
    assert torch.equal(planar[:, :, 4:53].permute(0, 2, 1), x) and planar[:, :, :4].abs().max() == 0


@pytest.mark.parametrize("fname,name,width", [("tcresnet8_1.0_4020.npz", "TCResNet8", 1.0), ("tcresnet8_1.0_3010.npz", "TCResNet8", 1.0),
                                              ("tcresnet14_1.5_4020.npz", "TCResNet14", 1.5), ("tcresnet14_1.5_3010.npz", "TCResNet14", 1.5)])
def test_eval_forward(hip_lib, fname, name, width):
    Cm.check_eval(hip_lib, fname, name, width)


@pytest.mark.parametrize("fname,name,width,steps", [("tcresnet8_1.0_4020.npz", "TCResNet8", 1.0, 3), ("tcresnet8_1.0_3010.npz", "TCResNet8", 1.0, 1),
                                                    ("tcresnet14_1.5_4020.npz", "TCResNet14", 1.5, 1), ("tcresnet14_1.5_3010.npz", "TCResNet14", 1.5, 1)])
def test_train(hip_lib, fname, name, width, steps):
    Cm.check_train(hip_lib, fname, name, width, steps=steps)


def _full_batch(hip_lib, tag, b=4096):
    win, hop = (640, 320) if tag == "4020" else (480, 160)
    arch = R.make_tcresnet("TCResNet8", 1.0)
    p, s = R.init_params(arch, 11)
    R.randomize_bn(arch, p, s)
    base = R.synth_waveforms(64, seed=5)
    wav = torch.from_numpy(np.tile(base, (b // 64, 1))).cuda()
    fe = Cm.make_frontend(hip_lib, win, hop)
    net = Cm.make_net(hip_lib, "TCResNet8", 1.0, fe.n_frames, p, s)
    return arch, p, s, base, wav, fe, net


@pytest.mark.parametrize("tag", ["4020", "3010"])
def test_full_batch_eval_properties(hip_lib, tag):
    """B = 4096 (BASELINE.json configs[1]): utterance independence (64 distinct utterances tiled 64x must give
    64 bit-identical copies), parity of the distinct rows with the oracle, softmax rows sum to 1."""
    arch, p, s, base, wav, fe, net = _full_batch(hip_lib, tag)
    logits, probs = net.forward_infer(fe(wav))
    l = logits.view(64, 64, 12)
    assert torch.equal(l, l[:1].expand_as(l)), "utterances in a batch are not independent"
    cfg = Cm.frontend_cfg(fe.cfg.win, fe.cfg.hop)
    ref = R.forward(arch, p, s, R.mfcc(base, cfg), False)
    assert np.abs(l[0].cpu().numpy() - ref["logits"]).max() < Cm.LOGIT_TOL
    assert np.array_equal(l[0].cpu().numpy().argmax(1), ref["logits"].argmax(1))
    assert torch.allclose(probs.sum(1), torch.ones(4096, device="cuda"), atol=1e-5)
    # permutation equivariance, bit exact
    perm = torch.randperm(4096, device="cuda")
    l2, _ = net.forward_infer(fe(wav[perm].contiguous()))
    assert torch.equal(l2, logits[perm])


def test_full_batch_train_properties(hip_lib):
    """B = 4096 training step: bitwise run-to-run determinism (no float atomics on the path) and agreement of
    the batch statistics / gradients with the oracle evaluated on the 64 distinct utterances (tiling a batch
    leaves mean/variance and the mean-loss gradient unchanged)."""
    arch, p, s, base, wav, fe, net = _full_batch(hip_lib, "4020")
    lab = torch.from_numpy(np.tile(R.synth_labels(64), (64, 1))).cuda()
    feat = fe(wav)
    outs = []
    for _ in range(2):
        sd = dict(p); sd.update(s); net.load_state_dict(sd)
        logits, probs, loss = net.forward_train(feat, lab, keep_prob=1.0)
        g = net.backward().clone()
        outs.append((logits.clone(), loss.clone(), g, net.stats.clone()))
    assert all(torch.equal(a, b) for a, b in zip(outs[0], outs[1])), "training step is not bitwise deterministic"
    x = R.mfcc(base, R.FRONTEND_4020)
    ref = R.forward(arch, p, s, x, True)
    assert np.abs(outs[0][0][:64].cpu().numpy() - ref["logits"]).max() < Cm.LOGIT_TOL          # north_star: 1e-4
    tot, model, _ = R.loss(ref["logits"], R.synth_labels(64).astype(np.float64), p, 0.0)
    assert abs(float(outs[0][1]) / 4096 - model) < 1e-4
    rg = R.backward(arch, p, ref, R.synth_labels(64).astype(np.float64), 0.0)
    # 310 k ReLU inputs: those within 1e-5 of the kink (counted from the float64 forward) may be masked differently in f32, and each
    # flip moves a gradient entry; none -> 2e-4 of the tensor's largest entry, otherwise 5e-4
    near = Cm.relu_near(arch, ref)
    tol = 2e-4 if near == 0 else 5e-4
    for k, v in rg.items():
        got = net.grad_view(k).cpu().numpy().reshape(v.shape)
        assert np.abs(got - v).max() < tol * max(np.abs(v).max(), 1e-3), (k, near)
    # moving variance uses the Bessel factor n/(n-1) of the ACTUAL count (4096*T), not of the 64-utterance oracle
    k = "TCResNet8/conv0/BatchNorm/moving_mean"
    assert np.abs(net._view(k).cpu().numpy() - ref["new_stats"][k]).max() < 1e-5


@pytest.mark.parametrize("size", ["S", "M", "L"])
def test_dscnn_eval_forward(hip_lib, size):
    Cm.check_dscnn(hip_lib, size)


def test_dscnn_full_batch_independence(hip_lib):
    """DSCNN-L at batch 4096 (BASELINE.json configs[4]): 64 distinct utterances tiled 64x give identical rows."""
    from oracle import dscnn_ref as D
    import dataclasses
    p, s = D.init_params(D.net_def("L"), seed=0)
    base = R.synth_waveforms(64, seed=9)
    fe = Cm.make_frontend(hip_lib, 640, 320, num_mfccs=10)
    net = T.DSCNN("L", fe.n_frames, 10, 12, device="cuda")
    sd = dict(p); sd.update(s); net.load_state_dict(sd)
    logits, probs = net.forward_infer(fe(torch.from_numpy(np.tile(base, (64, 1))).cuda()))
    l = logits.view(64, 64, 12)
    assert torch.equal(l, l[:1].expand_as(l))
    ref = D.forward(D.net_def("L"), p, s, R.mfcc(base, dataclasses.replace(R.FRONTEND_4020, num_mfccs=10)), False)
    assert np.abs(l[0].cpu().numpy() - ref["logits"]).max() < Cm.LOGIT_TOL


@pytest.mark.parametrize("size", ["M", "L"])
def test_dscnn_pointwise_kernel_paths_agree(hip_lib, size):
    """The LDS-tiled pointwise kernel (default for 97..288 output channels) and the register-fed 1x1 kernel
    (TCR_TUNE_CONV_B = 3; the knob also un-fuses conv_1 from the first depthwise layer) accumulate in the same order:
    bit-identical logits, eval mode, at a
    batch whose last workgroup is partial (97 utterances x 65 positions = 98.5 workgroups of 64 positions)."""
    from oracle import dscnn_ref as D
    p, s = D.init_params(D.net_def(size), seed=3)
    fe = Cm.make_frontend(hip_lib, 640, 320, num_mfccs=10)
    net = T.DSCNN(size, fe.n_frames, 10, 12, device="cuda")
    sd = dict(p); sd.update(s); net.load_state_dict(sd)
    feat = fe(torch.from_numpy(R.synth_waveforms(97, seed=5)).cuda())
    try:
        base = net.forward_infer(feat)[0].clone()
        hip_lib.tcr_tune(2, 3)
        alt = net.forward_infer(feat)[0].clone()
    finally:
        hip_lib.tcr_tune(2, 0)
    assert torch.isfinite(base).all()
    assert torch.equal(base, alt)


@pytest.mark.parametrize("size", ["S", "L"])
def test_dscnn_train_steps(hip_lib, size):
    """DS-CNN train-mode forward + backward + Adam against the oracle fixture (BASELINE.json configs[4], training half)."""
    Cm.check_dscnn_train(hip_lib, size, steps=3)


def test_dscnn_m_train_against_live_oracle(hip_lib):
    """DS-CNN-M (172 channels: the 96-row blocks of the LDS-tiled pointwise wgrad end in a partial block)."""
    Cm.check_dscnn_train_live(hip_lib, "M")


def test_dscnn_train_full_batch(hip_lib):
    """DSCNN-L training step at batch 4096: 64 distinct utterances tiled 64x have the batch statistics of the 64, so
    logits / loss equal the oracle's on the 64, every tile gets identical rows, gradients agree, and the step is
    bitwise reproducible."""
    from oracle import dscnn_ref as D
    import dataclasses
    blocks = D.net_def("L")
    p, s = D.init_params(blocks, seed=0)
    base = R.synth_waveforms(64, seed=9)
    labels64 = R.synth_labels(64).astype(np.float64)
    fe = Cm.make_frontend(hip_lib, 640, 320, num_mfccs=10)
    net = T.DSCNN("L", fe.n_frames, 10, 12, device="cuda")
    sd = dict(p); sd.update(s); net.load_state_dict(sd)
    feat = fe(torch.from_numpy(np.tile(base, (64, 1))).cuda())
    labels = torch.from_numpy(np.tile(labels64, (64, 1)).astype(np.float32)).cuda()
    stats0 = net.stats.clone()
    logits, probs, loss_sum = net.forward_train(feat, labels)
    g1 = net.backward().clone()
    l = logits.view(64, 64, 12)
    assert torch.equal(l, l[:1].expand_as(l))
    ref = D.forward(blocks, p, s, R.mfcc(base, dataclasses.replace(R.FRONTEND_4020, num_mfccs=10)), True)
    assert np.abs(l[0].cpu().numpy() - ref["logits"]).max() < Cm.LOGIT_TOL
    assert abs(float(loss_sum) / 4096 - D.loss(ref["logits"], labels64)) < 1e-4
    # 64 utterances x 11 BN+ReLU layers x 276 channels x 65-250 positions = ~20 M ReLU inputs, some tens of them within 1e-5 of the
    # kink, where an f32 forward may land on the other side; ONE flip moves a depthwise / pointwise gradient entry by O(1 / positions).
    # As for the graph-engine families (tests/test_models2d.py), the float64 oracle takes the kernels' own side for exactly those
    # inputs (tcr_dscnn_unit_output, consulted only where |xhat + beta| < 1e-5): every gradient then holds 1e-3 of its tensor's
    # largest entry at the full batch (round 2 allowed 2e-2 here).
    keys, near, masks = [], 0, {}
    for blk in blocks:
        keys += [(blk.scope + "/out", f"DSCNN/{blk.scope}/batch_norm")] if blk.type != "separable" else \
            [(blk.scope + "/mid", f"DSCNN/{blk.scope}/dw_batch_norm"), (blk.scope + "/out", f"DSCNN/{blk.scope}/pw_batch_norm")]
    assert len(keys) == net.lib.tcr_dscnn_num_units(net._h)
    for ui, (ck, bn) in enumerate(keys):
        pre = ref["cache"][bn]["xhat"] + p[bn + "/beta"]                    # [64, H, W, C]: the ReLU's input
        close = np.abs(pre) < 1e-5
        near += int(close.sum())
        if close.any():
            act = net.unit_output(ui, 4096)[:64].cpu().numpy()                # [64, C, H*W] (tile 0 of the batch)
            kpos = np.transpose(act.reshape(64, pre.shape[3], pre.shape[1], pre.shape[2]), (0, 2, 3, 1)) > 0
            masks[ck] = np.where(close, kpos, pre > 0)
    gref = D.backward(blocks, p, ref, labels64, masks=masks)
    # EVERY gradient tensor of the net (weights, depthwise weights, biases, BN betas: 35), not a hand-picked few.  Conv biases ahead of a
    # train-mode BN have an exactly-zero gradient (TF computes round-off noise there): an absolute bound relative to the net's largest entry.
    gmax = max(float(np.abs(v).max()) for v in gref.values())
    checked = 0
    for k in [n for n, ti in net.tensors.items() if ti.arena == 0]:
        if k not in gref:
            continue
        got = net.grad_view(k).cpu().numpy().reshape(gref[k].shape)
        scale = float(np.abs(gref[k]).max())
        tol = 1e-3 * scale if scale > 1e-6 * gmax else 1e-6 * gmax
        assert np.abs(got - gref[k]).max() <= tol, (k, float(np.abs(got - gref[k]).max()), scale, near)
        checked += 1
    assert checked == len(gref) >= 30, (checked, len(gref))          # DS-CNN-L: 35 gradient tensors
    net.stats.copy_(stats0)
    logits2, _, loss2 = net.forward_train(feat, labels)
    g2 = net.backward()
    assert torch.equal(logits, logits2) and torch.equal(g1, g2) and float(loss_sum) == float(loss2)


def test_every_kernel_path_agrees(hip_lib):
    """The alternative kernels behind the tcr_tune knobs (scalar-fed VALU convs, LDS-image MFMA convs, per-layer vs
    whole-network fused eval kernel, front-end variants) all reproduce the default path: bit-exact where the
    arithmetic order is identical (MFMA variants), within the logit tolerance otherwise."""
    fx = Cm.load("tcresnet8_1.0_4020.npz")
    arch, p, s = Cm.fixture_params(fx, "TCResNet8", 1.0)
    fe = Cm.make_frontend(hip_lib, fx["win"], fx["hop"])
    wav = Cm.to_dev(hip_lib, np.tile(fx["wav"], (40, 1)))          # 160 utterances: several workgroups / groups
    labels = Cm.to_dev(hip_lib, np.tile(fx["labels"], (40, 1)))
    net = Cm.make_net(hip_lib, "TCResNet8", 1.0, fe.n_frames, p, s)
    try:
        feat0 = fe(wav).clone()                                 # default: packed-FP32 kernel
        hip_lib.tcr_tune(1, 1)
        feat1 = fe(wav).clone()                                 # scalar-FP32 kernel
        assert (feat1 - feat0).abs().max() < 2e-4               # (different rounding order in the real-FFT split)
        for v in (2, 3, 4):                                     # its scheduling variants are bit-exact among themselves
            hip_lib.tcr_tune(1, v)
            assert torch.equal(fe(wav), feat1), f"front-end variant {v}"
        hip_lib.tcr_tune(1, 5)
        assert torch.equal(fe(wav), feat0)
        hip_lib.tcr_tune(1, 0)
        base, _ = net.forward_infer(feat0)
        assert np.abs(base[:4].cpu().numpy() - fx["eval_logits"]).max() < Cm.LOGIT_TOL
        results = {}
        for name, knobs in (("per-layer mfma", {3: 1}), ("per-layer mfma, LDS image", {3: 1, 2: 1}), ("per-layer valu", {3: 1, 0: 1}),
                            ("fused g=3", {4: 3}), ("fused g=4, 4 waves", {4: 4, 5: 404}), ("fused, features staged in LDS", {3: 2}),
                            ("fused g=8, 16 waves, ring 8", {4: 8, 5: 816}), ("fused, generic layer walk (no compile-time shapes)", {3: 3}),
                            ("fused, generic walk, 4 waves", {3: 3, 5: 404}),
                            ("fused, utterance strides not padded to the bank pattern", {3: 7}),
                            ("fused, four tiles per job in block 0's layers", {3: 5}),
                            ("fused, four tiles per job in block 0, 3 utterances per group", {3: 5, 4: 3}),
                            ("fused, nine-tap layers as jobs of two tiles dealt round-robin (rounds 3-5)", {3: 8}),
                            ("fused, jobs of two tiles, 3 utterances per group", {3: 8, 4: 3}),
                            ("fused, units without the weight lookahead", {3: 9}), ("fused, units without lookahead, 5 utterances, 4 waves", {3: 9, 4: 5, 5: 404}),
                            ("fused, 5 utterances, 4 waves", {4: 5, 5: 404}), ("fused, 7 utterances", {4: 7})):
            for k, v in knobs.items():
                hip_lib.tcr_tune(k, v)
            results[name] = net.forward_infer(feat0)[0].clone()
            for k in knobs:
                hip_lib.tcr_tune(k, 0)
        for name, r in results.items():
            if "valu" in name:
                assert (r - base).abs().max() < 2e-5, name
            else:
                assert torch.equal(r, base), name
        # training: MFMA vs VALU conv/dgrad paths give the same gradients up to f32 re-association
        grads = []
        for path, ksplit in ((0, 0), (1, 0), (0, 1), (0, 2), (0, 4)):
            hip_lib.tcr_tune(0, path)
            hip_lib.tcr_tune(6, ksplit)
            sd = dict(p); sd.update(s); net.load_state_dict(sd)
            net.forward_train(feat0, labels, keep_prob=0.5, seed=7)
            grads.append(net.backward().clone())
        hip_lib.tcr_tune(0, 0)
        hip_lib.tcr_tune(6, 0)
        for g in grads[1:]:
            assert (grads[0] - g).abs().max() < 2e-4 * max(1.0, float(grads[0].abs().max()))
    finally:
        for k in range(7):
            hip_lib.tcr_tune(k, 0)


@pytest.mark.parametrize("name,width,batch,t", [("TCResNet8", 1.0, 4096, 49), ("TCResNet8", 1.0, 131, 98), ("TCResNet8", 1.0, 517, 49),
                                                ("TCResNet14", 1.5, 203, 49), ("TCResNet14", 1.5, 1024, 98)])
def test_fused_eval_job_forms_are_bitwise(hip_lib, name, width, batch, t):
    Cm.check_fused_eval_job_forms(hip_lib, name, width, batch, t)


def test_feature_prefetch_matches_sequential(hip_lib):
    """FeaturePrefetcher (front-end of step k+1 on a second stream) leaves a training run bitwise unchanged."""
    from tcresnet_amd.pipeline import FeaturePrefetcher
    fx = Cm.load("tcresnet8_1.0_4020.npz")
    arch, p, s = Cm.fixture_params(fx, "TCResNet8", 1.0)
    fe = Cm.make_frontend(hip_lib, fx["win"], fx["hop"])
    batches = [Cm.to_dev(hip_lib, np.tile(np.roll(fx["wav"], k, axis=0), (32, 1))) for k in range(4)]
    labels = Cm.to_dev(hip_lib, np.tile(fx["labels"], (32, 1)))
    finals = []
    for prefetch in (False, True):
        net = Cm.make_net(hip_lib, "TCResNet8", 1.0, fe.n_frames, p, s)
        pf = FeaturePrefetcher(fe, batches[0].shape[0]) if prefetch else None
        if pf:
            pf.submit(batches[0])
        for k in range(4):
            if pf:
                feat = pf.get()
                if k + 1 < 4:
                    pf.submit(batches[k + 1])
            else:
                feat = fe(batches[k])
            net.forward_train(feat, labels, keep_prob=0.5, seed=k)
            net.backward()
            net.sgd_momentum_step(0.1, 0.9, 0.001)
        torch.cuda.synchronize()
        finals.append((net.params.clone(), net.stats.clone()))
    assert torch.equal(finals[0][0], finals[1][0]) and torch.equal(finals[0][1], finals[1][1])


@pytest.mark.parametrize("batch", [1, 2])
def test_small_batches_against_live_oracle(hip_lib, batch):
    """BASELINE.json configs[0] on the HIP path: batch = 1 (and 2) eval forward + a training step against the oracle."""
    Cm.check_small_batch(hip_lib, batch)
    Cm.check_small_batch(hip_lib, batch, tag="3010")


@pytest.mark.parametrize("name,width,batch", [("TCResNet8", 1.0, 4096), ("TCResNet14", 1.5, 4096), ("TCResNet14", 1.5, 37)])
def test_staged_sync_bn_api_is_bitwise_the_unstaged_path(hip_lib, name, width, batch):
    """configs[3]'s per-replica work (TCResNet14-1.5 at 4096 utterances) through the staged sync-BN API, world = 1."""
    Cm.check_staged_equals_unstaged(hip_lib, name, width, batch)


def test_wide_net_training(hip_lib):
    """width_multiplier 2.0 (Cout = 96 > 80): the shortcut's weight gradient is reduced on the main stream while its BN
    backward ran on the side stream -- a missing stream dependency shows up as wrong or run-to-run different gradients."""
    Cm.check_small_batch(hip_lib, 3, name="TCResNet8", width=2.0)
    Cm.check_small_batch(hip_lib, 3, name="TCResNet14", width=2.0)
    arch = R.make_tcresnet("TCResNet8", 2.0)
    p, s = R.init_params(arch, 1)
    fe = Cm.make_frontend(hip_lib, 640, 320)
    feat = fe(torch.from_numpy(np.tile(R.synth_waveforms(64, seed=3), (16, 1))).cuda())
    lab = torch.from_numpy(np.tile(R.synth_labels(64), (16, 1))).cuda()
    grads = []
    for _ in range(4):
        net = Cm.make_net(hip_lib, "TCResNet8", 2.0, fe.n_frames, p, s)
        net.forward_train(feat, lab, keep_prob=0.5, seed=1)
        grads.append(net.backward().clone())
    assert all(torch.equal(grads[0], g) for g in grads[1:])


def test_train_paths_agree(hip_lib):
    """Train-mode forward as group-resident phases (default) vs per-layer kernels (TCR_TUNE_TRAIN_FWD = 1); backward with the lazy BN
    backward (bwd_lazy.hip: default for nets of <= 48 channels, knob 3 for every net -- TCResNet14-1.5's identity shortcuts and 3-5 row
    tiles), the group-resident phases (1) and the per-layer chain (2): same logits / gradients up to f32 re-association, each path
    against the oracle fixture, each bitwise reproducible."""
    fx = Cm.load("tcresnet8_1.0_4020.npz")
    arch, p, s = Cm.fixture_params(fx, "TCResNet8", 1.0)
    fe = Cm.make_frontend(hip_lib, fx["win"], fx["hop"])
    feat = fe(Cm.to_dev(hip_lib, np.tile(fx["wav"], (256, 1))))
    labels = Cm.to_dev(hip_lib, np.tile(fx["labels"], (256, 1)))
    outs = {}
    try:
        for fwd, bwd in ((0, 0), (1, 0), (0, 1), (1, 1), (0, 2), (0, 3), (1, 3)):     # backward: 0 lazy (narrow nets) / 1 phases / 2 per-layer chain / 3 lazy for every net
            hip_lib.tcr_tune(8, fwd); hip_lib.tcr_tune(9, bwd)
            Cm.check_train(hip_lib, "tcresnet8_1.0_4020.npz", "TCResNet8", 1.0, steps=1)
            Cm.check_train(hip_lib, "tcresnet14_1.5_4020.npz", "TCResNet14", 1.5, steps=1)
            runs = []
            for _ in range(2):
                net = Cm.make_net(hip_lib, "TCResNet8", 1.0, fe.n_frames, p, s)
                lg, _, loss = net.forward_train(feat, labels, keep_prob=0.5, seed=4)
                runs.append((lg.clone(), net.backward().clone(), net.stats.clone()))
            assert all(torch.equal(a, b) for a, b in zip(*runs)), (fwd, bwd)
            outs[(fwd, bwd)] = runs[0]
    finally:
        hip_lib.tcr_tune(8, 0); hip_lib.tcr_tune(9, 0)
    ref = outs[(0, 0)]
    for k, o in outs.items():
        assert (o[0] - ref[0]).abs().max() < 1e-4 and (o[1] - ref[1]).abs().max() < 2e-4 * max(1.0, float(ref[1].abs().max())), k


def test_bn_backward_fused_equals_pair(hip_lib):
    Cm.check_bn_backward_fused_equals_pair(hip_lib, "TCResNet8", 1.0, 1024)
    Cm.check_bn_backward_fused_equals_pair(hip_lib, "TCResNet14", 1.5, 300)
    Cm.check_bn_backward_fused_equals_pair(hip_lib, "TCResNet8", 2.0, 64)       # 96 channels: the non-deferrable weight-gradient branch
    Cm.check_dscnn_mask_paths_agree(hip_lib, "S", 200)
    Cm.check_dscnn_mask_paths_agree(hip_lib, "L", 256)


@pytest.mark.gpu
def test_down_dgrad_order_is_bitwise(hip_lib):
    Cm.check_down_dgrad_order(hip_lib, "TCResNet8", 1.0, 4096, t=49)
    Cm.check_down_dgrad_order(hip_lib, "TCResNet14", 1.5, 300)
    Cm.check_down_dgrad_order(hip_lib, "TCResNet8", 1.5, 512, t=98)


@pytest.mark.parametrize("size,batch", [("S", 96), ("M", 100), ("L", 1024)])
def test_dscnn_staged_sync_bn_api(hip_lib, size, batch):
    Cm.check_dscnn_staged_equals_unstaged(hip_lib, size, batch)


@pytest.mark.parametrize("size,batch", [("L", 37), ("M", 256)])
def test_dscnn_pointwise_kernel_geometries_agree(hip_lib, size, batch):
    Cm.check_dscnn_pointwise_geometries(hip_lib, size, batch)


@pytest.mark.parametrize("size,batch", [("L", 37), ("M", 256)])
def test_dscnn_bn_backward_apply_kernels_agree(hip_lib, size, batch):
    Cm.check_dscnn_pointwise_geometries(hip_lib, size, batch, knob_id=29)


@pytest.mark.parametrize("size,batch", [("L", 37), ("M", 256)])
def test_dscnn_pointwise_filter_gradient_unrolled_kernel_is_bitwise(hip_lib, size, batch):
    Cm.check_dscnn_pointwise_geometries(hip_lib, size, batch, knob_id=25, alt=2)


@pytest.mark.parametrize("size,batch", [("L", 36), ("M", 256), ("L", 37)])         # (the row kernel takes whole blocks of 16 planes; 37 x 276 planes: its fallback, row and image kernels mixed within one step)
def test_dscnn_depthwise_forward_kernels_agree(hip_lib, size, batch):
    Cm.check_dscnn_pointwise_geometries(hip_lib, size, batch, knob_id=32)


@pytest.mark.parametrize("size,batch", [("L", 37), ("M", 256)])
def test_dscnn_depthwise_filter_gradient_kernels_agree(hip_lib, size, batch):
    Cm.check_dscnn_pointwise_wgrad_kernels(hip_lib, size, batch, knob_id=31)


@pytest.mark.parametrize("size,batch", [("L", 36), ("M", 256), ("L", 37)])         # (37 x 276 planes are not whole blocks of 16: the fallback dispatch)
def test_dscnn_depthwise_data_gradient_kernels_agree(hip_lib, size, batch):
    Cm.check_dscnn_pointwise_geometries(hip_lib, size, batch, knob_id=30)


@pytest.mark.parametrize("size,batch", [("L", 37), ("M", 256)])
def test_dscnn_pointwise_filter_gradient_kernels_agree(hip_lib, size, batch):
    Cm.check_dscnn_pointwise_wgrad_kernels(hip_lib, size, batch)


@pytest.mark.parametrize("size,batch", [("M", 67), ("L", 256), ("L", 4096)])
def test_dscnn_lazy_training_path_equals_materialised(hip_lib, size, batch):
    """Default DS-CNN training (BN + ReLU applied by the consumers, sums from the conv / data-gradient epilogues; ragged and full batches)
    against the materialising path behind TCR_TUNE_DS_TRAIN = 1."""
    Cm.check_dscnn_lazy_equals_materialised(hip_lib, size, batch)


def test_forward_waveform_single_call_is_the_three_call_path(hip_lib):
    """tcr_forward_waveform (front-end + fold-if-stale + network in one host call) == frontend -> forward_infer, bitwise; the fold
    follows weight changes (load_state_dict, an optimiser step)."""
    lib = hip_lib
    fx = Cm.load("tcresnet8_1.0_4020.npz")
    arch, p, s = Cm.fixture_params(fx, "TCResNet8", 1.0)
    fe = Cm.make_frontend(lib, fx["win"], fx["hop"])
    net = Cm.make_net(lib, "TCResNet8", 1.0, fe.n_frames, p, s)
    wav = Cm.to_dev(lib, fx["wav"])
    ref = net.forward_infer(fe(wav), want_ranges=True)
    got = net.forward_waveform(fe, wav, want_ranges=True)
    for a, b in zip(got, ref):
        assert torch.equal(a, b)
    assert np.abs(got[0].cpu().numpy() - fx["eval_logits"]).max() < Cm.LOGIT_TOL
    # weights change -> the cached fold is refreshed inside the same call
    s2 = {k: (v * 1.5 if "moving_variance" in k else v + 0.1) for k, v in s.items()}
    sd = dict(p); sd.update(s2)
    net.load_state_dict(sd)
    got2 = net.forward_waveform(fe, wav[:3].contiguous())
    fresh = Cm.make_net(lib, "TCResNet8", 1.0, fe.n_frames, p, s2)
    ref2 = fresh.forward_infer(fe(wav[:3].contiguous()))
    assert torch.equal(got2[0], ref2[0]) and torch.equal(got2[1], ref2[1]) and not torch.equal(got2[0], got[0][:3])
    lab = Cm.to_dev(lib, fx["labels"])
    net.forward_train(fe(wav), lab, keep_prob=1.0); net.backward(); net.sgd_momentum_step(0.1, 0.9, 0.001)
    got3 = net.forward_waveform(fe, wav)
    ref3 = net.forward_infer(fe(wav))
    assert torch.equal(got3[0], ref3[0]) and not torch.equal(got3[0], got[0])


@pytest.mark.parametrize("family", ["TCResNet8", "TCResNet14", "DSCNN-L"])
def test_multi_stream_backward_repeats_bitwise(hip_lib, family):
    """The backward runs on up to three internal streams (filter gradients, the shortcut units' chains, the classifier's gradient) with
    rotating gradient buffers; a race would show up as a run-to-run difference: eight repetitions at batch 4096 are bitwise equal."""
    B = 4096
    rng = np.random.RandomState(11)
    labels = torch.from_numpy(R.synth_labels(B).astype(np.float32)).cuda()
    if family == "DSCNN-L":
        feat = T.features_to_planar(torch.from_numpy(rng.uniform(-2, 2, (B, 49, 10)).astype(np.float32)).cuda(), lib=hip_lib)
        net = T.DSCNN("L", 49, 10, 12, device="cuda")
    else:
        feat = T.features_to_planar(torch.from_numpy(rng.uniform(-2, 2, (B, 49, 40)).astype(np.float32)).cuda(), lib=hip_lib)
        net = T.TCResNet(family, R.tcresnet_channels(family, 1.0 if family == "TCResNet8" else 1.5), 40, 49, 12, device="cuda")
    net.init_xavier(3)
    stats0 = net.stats.clone()
    ref = None
    for _ in range(8):
        net.stats.copy_(stats0)
        lg, _, _ = net.forward_train(feat, labels, keep_prob=0.5, seed=5)
        g = net.backward().clone()
        if ref is None:
            ref = (lg.clone(), g)
        else:
            assert torch.equal(lg, ref[0]) and torch.equal(g, ref[1])


@pytest.mark.gpu
@pytest.mark.parametrize("mode,depth", [("handoff", 2), ("handoff", 3), ("alternate", 2), ("alternate", 3), ("alternate", 4)])
@pytest.mark.parametrize("width,batch", [(1.0, 2048), (1.5, 96)])
def test_inference_pipeline_equals_sequential(hip_lib, mode, depth, width, batch):
    """The two-stream inference pipelines (tcresnet_amd.pipeline.InferencePipeline: front-end stream -> network stream, or whole batches
    alternating between the streams) return, for a sequence of DIFFERENT batches, bitwise what front-end -> network gives on one stream;
    width 1.5 at a small batch takes the per-layer eval path, whose activations live in the workspace (one per stream when alternating)."""
    from tcresnet_amd.pipeline import InferencePipeline
    fe = T.Frontend(window_size_samples=640, window_stride_samples=320, device="cuda")
    net = T.TCResNet("TCResNet8", R.tcresnet_channels("TCResNet8", width), 40, fe.n_frames, 12, device="cuda")
    net.init_xavier(7)
    base = torch.from_numpy(R.synth_waveforms(64, seed=21)).cuda()
    wavs = [(base.roll(k, 0) * (1.0 - 0.05 * k)).repeat((batch + 63) // 64, 1)[:batch].contiguous() for k in range(7)]
    want = [net.forward_infer(fe(w))[0].clone() for w in wavs]
    assert not torch.equal(want[0], want[1])
    pipe = InferencePipeline(fe, net, batch, depth=depth, mode=mode, ways=depth if mode == "alternate" else 2)      # (alternate: streams = batches in flight)
    assert pipe.depth == depth
    got = []
    for k, w in enumerate(wavs):
        out = pipe.submit(w)
        pipe.done_event(k % pipe.depth).synchronize()
        got.append(out[0].clone())
    pipe.sync()
    for g, r in zip(got, want):
        assert torch.equal(g, r)
    # back to back, no host synchronisation between submits: the last `depth` results are still in their slots
    for rep in range(3):
        for w in wavs:
            pipe.submit(w)
    pipe.sync()
    torch.cuda.synchronize()
    n = len(wavs) * 3
    for k in range(n - pipe.depth, n):
        assert torch.equal(pipe.out[(7 + k) % pipe.depth][0], want[k % 7])


@pytest.mark.gpu
@pytest.mark.parametrize("name,width,batch,t", [("TCResNet8", 1.0, 4096, 49), ("TCResNet14", 1.5, 1030, 49), ("TCResNet8", 1.0, 515, 98),
                                                ("TCResNet14", 1.5, 259, 98)])
def test_static_phase_kernels_are_bitwise_the_generic_walk(hip_lib, name, width, batch, t):
    Cm.check_phase_kernel_variants(hip_lib, name, width, batch, t)


@pytest.mark.gpu
@pytest.mark.parametrize("name,width,batch,t", [("TCResNet8", 1.0, 4096, 49), ("TCResNet14", 1.5, 1031, 98), ("TCResNet8", 1.0, 3, 49)])
def test_first_conv_filter_gradient_kernels_agree(hip_lib, name, width, batch, t):
    Cm.check_first_conv_wgrad_kernels_agree(hip_lib, name, width, batch, t)


@pytest.mark.gpu
def test_lazy_backward_staging_forms_are_bitwise(hip_lib):
    Cm.check_backward_knob_variants(hip_lib, 22, (1,), True, batch=4096)
    Cm.check_backward_knob_variants(hip_lib, 22, (1,), True, batch=517, t=98)


@pytest.mark.gpu
def test_filter_gradient_waves_per_workgroup_agree(hip_lib):
    Cm.check_backward_knob_variants(hip_lib, 20, (8, 12, 16), False, batch=4096)


@pytest.mark.gpu
@pytest.mark.parametrize("name,width,batch,t", [("TCResNet8", 1.0, 4096, 49), ("TCResNet14", 1.5, 4096, 49), ("TCResNet14", 1.5, 1031, 98), ("TCResNet8", 1.0, 5, 98)])
def test_pipelined_filter_gradient_is_bitwise(hip_lib, name, width, batch, t):
    Cm.check_backward_knob_variants(hip_lib, 33, (1,), True, name=name, width=width, batch=batch, t=t)


@pytest.mark.gpu
@pytest.mark.parametrize("name,width,batch", [("TCResNet8", 1.0, 4096), ("TCResNet14", 1.5, 4096), ("TCResNet14", 1.5, 37)])
def test_wide_bn_backward_apply_loop_is_bitwise(hip_lib, name, width, batch):
    for mask in (0, 1):
        try:
            hip_lib.tcr_tune(9, 2); hip_lib.tcr_tune(12, mask)
            Cm.check_backward_knob_variants(hip_lib, 29, (1,), True, name=name, width=width, batch=batch, t=49)
        finally:
            hip_lib.tcr_tune(9, 0); hip_lib.tcr_tune(12, 0)


@pytest.mark.gpu
def test_last_block_two_unit_reduction_is_bitwise(hip_lib):
    Cm.check_backward_knob_variants(hip_lib, 12, (5,), True, batch=4096)
    Cm.check_backward_knob_variants(hip_lib, 12, (5,), True, batch=517, t=98)


@pytest.mark.gpu
def test_head_from_the_closing_phase_sums_is_bitwise(hip_lib):
    Cm.check_backward_knob_variants(hip_lib, 8, (2,), True, batch=4096)
    Cm.check_backward_knob_variants(hip_lib, 8, (2,), True, name="TCResNet14", width=1.5, batch=1031, t=98)


@pytest.mark.gpu
def test_prepared_waveform_call(hip_lib):
    """TCResNet.waveform_call: the single C-ABI call with its pointers bound once (the latency regime) == forward_waveform, bitwise, call
    after call into the same buffers; it refuses to run once the weights it was prepared for have changed."""
    lib = hip_lib
    fx = Cm.load("tcresnet8_1.0_4020.npz")
    arch, p, s = Cm.fixture_params(fx, "TCResNet8", 1.0)
    fe = Cm.make_frontend(lib, fx["win"], fx["hop"])
    net = Cm.make_net(lib, "TCResNet8", 1.0, fe.n_frames, p, s)
    wav = Cm.to_dev(lib, fx["wav"][:1])
    ref = [t.clone() for t in net.forward_waveform(fe, wav)]
    out = (torch.zeros_like(ref[0]), torch.zeros_like(ref[1]))
    call = net.waveform_call(fe, wav, out)
    for _ in range(3):
        out[0].zero_(); out[1].zero_()
        lib.tcr_tune(0, 0)          # (any host work between calls)
        # zero_() bumps neither params nor stats: the prepared call stays valid
        call()
        assert torch.equal(out[0], ref[0]) and torch.equal(out[1], ref[1])
    assert np.abs(out[0].cpu().numpy() - fx["eval_logits"][:1]).max() < Cm.LOGIT_TOL
    lab = Cm.to_dev(lib, fx["labels"])
    net.forward_train(fe(Cm.to_dev(lib, fx["wav"])), lab, keep_prob=1.0); net.backward(); net.sgd_momentum_step(0.1, 0.9, 0.001)
    with pytest.raises(T.TcrError):
        call()
    call2 = net.waveform_call(fe, wav, out)
    call2()
    assert torch.equal(out[0], net.forward_waveform(fe, wav)[0]) and not torch.equal(out[0], ref[0])
