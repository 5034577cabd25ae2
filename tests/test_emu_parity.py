"""CPU: the kernel SOURCES (tc-resnet_amd/csrc) run under the host-side wave-64 emulator (tests/emu) and are
checked against the golden vectors -- host logic + kernel logic without a GPU.  The gfx950 build of the same
sources is checked by tests/test_gpu_parity.py (-m gpu)."""
import numpy as np
import pytest
import torch

import tcresnet_amd as T
from oracle import numpy_ref as R
from tests import common as Cm


@pytest.mark.parametrize("tag", ["3010", "4020"])
def test_frontend(emu_lib, tag):
    Cm.check_frontend(emu_lib, tag)


@pytest.mark.parametrize("tag", ["3010", "4020"])
def test_frontend_edge_rows(emu_lib, tag):
    Cm.check_frontend_edges(emu_lib, tag)


def test_small_batch_network_kernel_is_bitwise_the_throughput_kernel(emu_lib):
    Cm.check_small_batch_kernel(emu_lib, batches=(1, 3))


@pytest.mark.parametrize("tag", ["3010", "4020"])
def test_pure_tone_rows_are_where_a_float32_fft_is(emu_lib, tag):
    """The kernels' error on the pure-tone fixture rows against what a complete float32 NumPy pipeline with a single-precision FFT
    (scipy.fft / pocketfft) makes of the same rows: the same class (within 2x), i.e. the 1e-3 bound of those rows is float32's, not the
    radix-16 decomposition's (tests/test_sensitivity.py::test_pure_tone_rows_what_float32_can_hold)."""
    import scipy.fft
    fx = Cm.load(f"frontend_edge_{tag}.npz")
    cfg = Cm.frontend_cfg(fx["win"], fx["hop"])
    fe = Cm.make_frontend(emu_lib, fx["win"], fx["hop"])
    got = fe.reference_view(fe(torch.from_numpy(fx["wav"])))[..., 0].numpy()
    e_kernel = np.abs(got - fx["mfcc"]).reshape(6, -1).max(1)
    frames = (R.frame_signal(fx["wav"].astype(np.float32), cfg.win, cfg.hop) * R.hann_periodic(cfg.win, np.float32)).astype(np.float32)
    x = np.zeros(frames.shape[:-1] + (cfg.nfft,), np.float32)
    x[..., :cfg.win] = frames
    X = scipy.fft.rfft(x, axis=-1)
    assert X.dtype == np.complex64
    power = (X.real ** 2 + X.imag ** 2).astype(np.float32)
    M = R.linear_to_mel_weight_matrix(cfg.num_mel_bins, cfg.n_bins, cfg.sample_rate, cfg.lower_edge_hertz, cfg.upper_edge_hertz).astype(np.float32)
    lm = np.log((power @ M).astype(np.float32) + np.float32(1e-6)).astype(np.float32)
    ref32 = (lm @ R.dct2_matrix(cfg.num_mel_bins, cfg.num_mfccs, np.float64).astype(np.float32)).astype(np.float32)
    e_f32 = np.abs(ref32 - fx["mfcc"]).reshape(6, -1).max(1)
    assert e_f32[4:].min() > 1.5e-4                                     # no float32 pipeline meets 2e-4 on these rows
    assert np.all(e_kernel[4:] < 2.0 * e_f32[4:] + 1e-5), (e_kernel, e_f32)
    assert np.all(e_kernel[:4] < 2.0 * e_f32[:4] + 2e-5), (e_kernel, e_f32)


@pytest.mark.parametrize("tag", ["3010", "4020"])
def test_edge_rows_logits(emu_lib, tag):
    Cm.check_edge_rows_logits(emu_lib, tag)


def test_frontend_three_wave_kernel_is_bitwise_the_two_wave_kernel(emu_lib):
    Cm.check_frontend_kernels_bitwise(emu_lib, 5)


def test_frontend_variants(emu_lib):
    fx = Cm.load("frontend_4020.npz")
    wav = torch.from_numpy(fx["wav"])
    # num_mfccs = 10 (the DS-CNN scripts) is a prefix of the 40-coefficient result
    f10 = Cm.make_frontend(emu_lib, 640, 320, num_mfccs=10)
    got = f10.reference_view(f10(wav))[..., 0].numpy()
    assert np.abs(got - fx["mfcc"][..., :10]).max() < Cm.MFCC_TOL
    # log_mel_spectrogram preprocessor: magnitude spectrum, no DCT (datasets/preprocessors.py:161-169)
    flm = Cm.make_frontend(emu_lib, 640, 320, method="log_mel_spectrogram")
    got = flm.reference_view(flm(wav))[..., 0].numpy()
    assert got.shape[-1] == 64 and np.abs(got[:2] - fx["log_mel_magnitude"][:2]).max() < 1e-4
    # [B, n, 1] input, ragged batch (not a multiple of the 64-frame workgroup tile), B = 1
    one = f10(wav[:1].unsqueeze(-1))
    assert np.array_equal(one.numpy(), f10(wav)[:1].numpy())
    with pytest.raises(T.TcrError):
        f10(wav[:, :100])
    with pytest.raises(NotImplementedError):
        Cm.make_frontend(emu_lib, 640, 320, method="spectrogram")


def test_no_preprocessing_relayout(emu_lib):
    x = torch.randn(3, 49, 40)
    planar = T.features_to_planar(x, lib=emu_lib)
    assert planar.shape == (3, 40, 57)
    assert torch.equal(planar[:, :, 4:53].permute(0, 2, 1), x) and planar[:, :, :4].abs().max() == 0 and planar[:, :, 53:].abs().max() == 0


@pytest.mark.parametrize("fname,name,width", [("tcresnet8_1.0_4020.npz", "TCResNet8", 1.0), ("tcresnet8_1.0_3010.npz", "TCResNet8", 1.0),
                                              ("tcresnet14_1.5_4020.npz", "TCResNet14", 1.5), ("tcresnet14_1.5_3010.npz", "TCResNet14", 1.5)])
def test_eval_forward(emu_lib, fname, name, width):
    Cm.check_eval(emu_lib, fname, name, width)


def test_train_three_steps(emu_lib):
    Cm.check_train(emu_lib, "tcresnet8_1.0_4020.npz", "TCResNet8", 1.0, steps=3)


def test_train_asymmetric_padding(emu_lib):
    Cm.check_train(emu_lib, "tcresnet8_1.0_3010.npz", "TCResNet8", 1.0, steps=1)     # T=98: SAME pads (3,4)


def test_train_identity_shortcuts_and_wide_channels(emu_lib):
    Cm.check_train(emu_lib, "tcresnet14_1.5_4020.npz", "TCResNet14", 1.5, steps=1)


def test_train_configs3_at_the_reference_setting(emu_lib):
    """TCResNet14-1.5 at 30 / 10 ms -> 98 frames, the reference's own script for this model (BASELINE configs[3];
    scripts/commands/TCResNet14Model-1.5_mfcc_40_3010_0.001_mom_l1.sh:3): asymmetric SAME pads (3, 4) in the 36- / 48- / 72-channel blocks."""
    Cm.check_train(emu_lib, "tcresnet14_1.5_3010.npz", "TCResNet14", 1.5, steps=1)


def test_batch_independence_in_eval(emu_lib):
    """Eval-mode utterances are independent: a batch of 70 (ragged vs the 64-position tiles) reproduces
    the single-utterance results bit for bit."""
    arch = R.make_tcresnet("TCResNet8", 1.0)
    p, s = R.init_params(arch, 3)
    R.randomize_bn(arch, p, s)
    wav = torch.from_numpy(R.synth_waveforms(70, seed=7))
    fe = Cm.make_frontend(emu_lib, 640, 320)
    net = Cm.make_net(emu_lib, "TCResNet8", 1.0, fe.n_frames, p, s)
    big, _ = net.forward_infer(fe(wav))
    for i in (0, 63, 64, 69):
        one, _ = net.forward_infer(fe(wav[i:i + 1]))
        assert torch.equal(one[0], big[i])


def test_engine_argument_errors(emu_lib):
    net = T.TCResNet("TCResNet8", [16, 24, 32, 48], 40, 49, 12, lib=emu_lib)
    with pytest.raises(T.TcrError):
        net.forward_infer(torch.zeros(2, 40, 49))            # not the padded planar layout
    with pytest.raises(T.TcrError):
        net.forward_infer(torch.zeros(2, 40, 57, dtype=torch.float64))
    with pytest.raises(T.TcrError):
        net.load_state_dict({n: np.zeros(3) for n in net.tensors})
    with pytest.raises(KeyError):
        net.load_state_dict({})
    net.init_xavier(0)
    sd = net.state_dict()
    assert sd["TCResNet8/conv0/weights"].shape == (3, 1, 40, 16) and abs(sd["TCResNet8/conv0/weights"]).max() <= np.sqrt(6.0 / (120 + 48)) + 1e-6
    assert np.all(sd["TCResNet8/conv0/BatchNorm/gamma"] == 1) and np.all(sd["TCResNet8/conv0/BatchNorm/moving_variance"] == 1)
    assert net.total_params() == 65264


@pytest.mark.parametrize("tag", ["3010", "4020"])
def test_frontend_deploy_path(emu_lib, tag):
    Cm.check_frontend_deploy(emu_lib, tag)


@pytest.mark.parametrize("size", ["S", "M", "L"])
def test_dscnn_eval_forward(emu_lib, size):
    Cm.check_dscnn(emu_lib, size)


@pytest.mark.parametrize("size", ["S", "L"])
def test_dscnn_train_steps(emu_lib, size):
    """Train-mode forward + backward + Adam (2 steps / 1 step) of DS-CNN; S: (2,2)/(1,1) strides, L: (2,1)/(2,2) and 276 channels."""
    Cm.check_dscnn_train(emu_lib, size, steps=2 if size == "S" else 1)


@pytest.mark.parametrize("batch", [1, 2])
def test_small_batches_against_live_oracle(emu_lib, batch):
    """BASELINE.json configs[0]: batch = 1 (and 2) eval forward + a training step."""
    Cm.check_small_batch(emu_lib, batch, seeds=range(100, 104))


def test_staged_sync_bn_api_is_bitwise_the_unstaged_path(emu_lib):
    Cm.check_staged_equals_unstaged(emu_lib, "TCResNet8", 1.0, batch=5)
    Cm.check_staged_equals_unstaged(emu_lib, "TCResNet8", 1.0, batch=5, handoff="unit")


def test_wide_net_training(emu_lib):
    """width_multiplier 2.0: 96 output channels take the non-deferrable weight-gradient branch."""
    Cm.check_small_batch(emu_lib, 3, name="TCResNet8", width=2.0, seeds=range(100, 103))


def test_backward_phases_opt_in(emu_lib):
    """TCR_TUNE_TRAIN_BWD = 1: the group-resident backward phases (an alternative kept behind the knob) against the fixtures, and
    through the staged sync-BN API bitwise like the unstaged run."""
    try:
        emu_lib.tcr_tune(9, 1)
        Cm.check_train(emu_lib, "tcresnet8_1.0_4020.npz", "TCResNet8", 1.0, steps=1)
        Cm.check_staged_equals_unstaged(emu_lib, "TCResNet8", 1.0, batch=3, handoff="unit", bwd_knob=1)     # (the level hand-off runs the per-layer backward)
    finally:
        emu_lib.tcr_tune(9, 0)


def test_backward_chains_against_the_fixtures(emu_lib):
    """TCR_TUNE_TRAIN_BWD: 3 = the lazy BN backward (bwd_lazy.hip) for every net -- TCResNet14-1.5 takes the per-layer chain by default:
    identity shortcuts, 36 / 48 / 72-channel gradients = three to five row tiles per job --, 2 = the per-layer chain for the narrow nets
    that default to the lazy one; each against the oracle fixtures, and the lazy one through the level hand-off bitwise the unstaged run."""
    try:
        emu_lib.tcr_tune(9, 3)
        Cm.check_train(emu_lib, "tcresnet14_1.5_4020.npz", "TCResNet14", 1.5, steps=1)
        Cm.check_staged_equals_unstaged(emu_lib, "TCResNet14", 1.5, batch=2, bwd_knob=3)
        emu_lib.tcr_tune(9, 2)
        Cm.check_train(emu_lib, "tcresnet8_1.0_4020.npz", "TCResNet8", 1.0, steps=1)
        Cm.check_train(emu_lib, "tcresnet8_1.0_3010.npz", "TCResNet8", 1.0, steps=1)
    finally:
        emu_lib.tcr_tune(9, 0)


def test_bn_backward_fused_equals_pair(emu_lib):
    Cm.check_bn_backward_fused_equals_pair(emu_lib, "TCResNet8", 1.0, 2, combos=((0, 0), (1, 1), (96, 0)))
    Cm.check_bn_backward_fused_equals_pair(emu_lib, "TCResNet14", 1.5, 1, combos=((96, 0), (1, 1)))    # 36 / 72 channels: channel blocks of 4 and 8
    Cm.check_dscnn_mask_paths_agree(emu_lib, "S", 2)


def test_down_dgrad_order_is_bitwise(emu_lib):
    Cm.check_down_dgrad_order(emu_lib, "TCResNet8", 1.0, 2, variants=((2, 0), (1, 0), (2, 2)))
    Cm.check_down_dgrad_order(emu_lib, "TCResNet14", 1.5, 1, t=24, variants=((2, 0), (1, 0)))     # even frame count: the other SAME-padding split


def test_dscnn_staged_sync_bn_api(emu_lib):
    Cm.check_dscnn_staged_equals_unstaged(emu_lib, "S", 2)
    Cm.check_dscnn_staged_equals_unstaged(emu_lib, "M", 2)      # the lazy path: hand-offs from the epilogue sums


def test_dscnn_pointwise_filter_gradient_kernels_agree(emu_lib):
    Cm.check_dscnn_pointwise_wgrad_kernels(emu_lib, "M", 3)


def test_dscnn_pointwise_kernel_geometries_agree(emu_lib):
    Cm.check_dscnn_pointwise_geometries(emu_lib, "M", 3)


def test_dscnn_bn_backward_apply_kernels_agree(emu_lib):
    Cm.check_dscnn_pointwise_geometries(emu_lib, "M", 3, knob_id=29)


def test_dscnn_pointwise_filter_gradient_unrolled_kernel_is_bitwise(emu_lib):
    Cm.check_dscnn_pointwise_geometries(emu_lib, "M", 3, knob_id=25, alt=2)


def test_dscnn_depthwise_forward_kernels_agree(emu_lib):
    Cm.check_dscnn_pointwise_geometries(emu_lib, "M", 4, knob_id=32)        # (whole blocks of 16 planes: 4 x 172)


def test_dscnn_depthwise_filter_gradient_kernels_agree(emu_lib):
    Cm.check_dscnn_pointwise_wgrad_kernels(emu_lib, "M", 3, knob_id=31)


def test_dscnn_depthwise_data_gradient_kernels_agree(emu_lib):
    Cm.check_dscnn_pointwise_geometries(emu_lib, "M", 4, knob_id=30)        # (the row kernel takes whole blocks of 16 planes: 4 x 172)


def test_dscnn_lazy_training_path_equals_materialised(emu_lib):
    Cm.check_dscnn_lazy_equals_materialised(emu_lib, "M", 2)


def test_forward_waveform_single_call_is_the_three_call_path(emu_lib):
    """tcr_forward_waveform (front-end + fold-if-stale + network in one host call) == frontend -> forward_infer, bitwise; the fold
    follows weight changes (load_state_dict, an optimiser step)."""
    lib = emu_lib
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


def test_prefetch_submit_point_policy(emu_lib):
    """FeaturePrefetcher.submit_point: every net family takes the next batch's front-end in front of its forward (measured, pipeline.py)."""
    from tcresnet_amd.pipeline import FeaturePrefetcher
    n8 = T.TCResNet("TCResNet8", [16, 24, 32, 48], 40, 49, 12, lib=emu_lib)
    n14 = T.TCResNet("TCResNet14", [24, 36, 36, 48, 48, 72, 72], 40, 49, 12, lib=emu_lib)
    assert FeaturePrefetcher.submit_point(n8) == "before_forward" and FeaturePrefetcher.submit_point(n14) == "before_forward"


@pytest.mark.parametrize("name,width,batch,t", [("TCResNet8", 1.0, 11, 49), ("TCResNet14", 1.5, 5, 49), ("TCResNet8", 1.0, 3, 98)])
def test_static_phase_kernels_are_bitwise_the_generic_walk(emu_lib, name, width, batch, t):
    Cm.check_phase_kernel_variants(emu_lib, name, width, batch, t)


@pytest.mark.parametrize("name,width,batch,t", [("TCResNet8", 1.0, 11, 49), ("TCResNet14", 1.5, 9, 49)])
def test_fused_eval_job_forms_are_bitwise(emu_lib, name, width, batch, t):
    Cm.check_fused_eval_job_forms(emu_lib, name, width, batch, t)


@pytest.mark.parametrize("name,width,batch,t", [("TCResNet8", 1.0, 13, 49), ("TCResNet14", 1.5, 6, 98)])
def test_first_conv_filter_gradient_kernels_agree(emu_lib, name, width, batch, t):
    Cm.check_first_conv_wgrad_kernels_agree(emu_lib, name, width, batch, t)


def test_lazy_backward_staging_forms_are_bitwise(emu_lib):
    Cm.check_backward_knob_variants(emu_lib, 22, (1,), True)


def test_filter_gradient_waves_per_workgroup_agree(emu_lib):
    Cm.check_backward_knob_variants(emu_lib, 20, (8, 16), False, batch=37)


@pytest.mark.parametrize("name,width,batch,t", [("TCResNet8", 1.0, 9, 49), ("TCResNet14", 1.5, 3, 49), ("TCResNet14", 1.5, 2, 98)])
def test_pipelined_filter_gradient_is_bitwise(emu_lib, name, width, batch, t):
    """conv_wgrad_mfma4_kernel with the next trip's operands requested ahead of the current trip's MFMAs (round 6; default) against the
    load-wait-multiply loop (TCR_TUNE_WGRAD_PIPE = 1): the same trips in the same order -- 16- and 8-position trips, rows of 7 / 13 /
    25 / 49 / 98 frames, a wave's last trip repeated as its dummy request, ragged utterance chunks -- gradients bitwise."""
    Cm.check_backward_knob_variants(emu_lib, 33, (1,), True, name=name, width=width, batch=batch, t=t)


@pytest.mark.parametrize("name,width,batch", [("TCResNet8", 1.0, 5), ("TCResNet14", 1.5, 3)])
def test_wide_bn_backward_apply_loop_is_bitwise(emu_lib, name, width, batch):
    """bn_bwd_apply_fused_kernel's wide loop (four float4 per thread and operand, coefficient rows from LDS; round 6) against its
    one-float4 loop (TCR_TUNE_BN_APPLY = 1), in the per-layer chain (TCR_TUNE_TRAIN_BWD = 2) where every unit takes it: bitwise dy,
    hence bitwise gradients; with the ReLU masks read back (TCR_TUNE_BWD_MASK = 1) the m1 / m2 operands ride along."""
    for mask in ((0, 1) if name == "TCResNet8" else (0,)):
        try:
            emu_lib.tcr_tune(9, 2); emu_lib.tcr_tune(12, mask)
            Cm.check_backward_knob_variants(emu_lib, 29, (1,), True, name=name, width=width, batch=batch, t=49)
        finally:
            emu_lib.tcr_tune(9, 0); emu_lib.tcr_tune(12, 0)


def test_last_block_two_unit_reduction_is_bitwise(emu_lib):
    """The lazy backward's first reduction: conv_b's and the shortcut's sums of the last block in one pass (chan_reduce_kernel<2>; round 6)
    against two launches, the second reading the masked gradient back (TCR_TUNE_BWD_MASK = 5): the same rows, gradients bitwise."""
    Cm.check_backward_knob_variants(emu_lib, 12, (5,), True, batch=9)
    Cm.check_backward_knob_variants(emu_lib, 12, (5,), True, batch=3, t=98)


def test_head_from_the_closing_phase_sums_is_bitwise(emu_lib):
    """The training forward's head starting from the sums over time that the closing phase leaves (round 6; default) against the head walking
    the block output's rows itself (TCR_TUNE_TRAIN_FWD = 2): the same frames added in the same order -- gradients bitwise."""
    Cm.check_backward_knob_variants(emu_lib, 8, (2,), True, batch=9)
    Cm.check_backward_knob_variants(emu_lib, 8, (2,), True, name="TCResNet14", width=1.5, batch=3, t=98)


def test_prepared_waveform_call(emu_lib):
    """TCResNet.waveform_call: the single C-ABI call with its pointers bound once (the latency regime) == forward_waveform, bitwise, call
    after call into the same buffers; it refuses to run once the weights it was prepared for have changed."""
    lib = emu_lib
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
