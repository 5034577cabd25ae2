"""Shared parity checks: the same assertions run against the emulator build (CPU, `-m "not gpu"`) and the
gfx950 library on a real MI355X (`-m gpu`).  The oracle is only ever the checker."""
import os

import numpy as np
import torch

import tcresnet_amd as T
from oracle import numpy_ref as R
from oracle.make_golden import dropout_mask

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
LOGIT_TOL = 1e-4          # north_star: logits within 1e-4 (fp32) of the reference CPU path
MFCC_TOL = 1e-4           # 1e-4 abs on MFCC (values reach ~ +-150)


def load(name):
    return dict(np.load(os.path.join(GOLDEN, name), allow_pickle=False))


def device_of(lib):
    return torch.device("cuda" if lib.kind == "hip" else "cpu")


def to_dev(lib, a):
    return torch.as_tensor(np.ascontiguousarray(a, dtype=np.float32)).to(device_of(lib))


def frontend_cfg(win, hop, num_mfccs=40):
    import dataclasses
    return dataclasses.replace(R.FRONTEND_3010, window_size_ms=win / 16.0, window_stride_ms=hop / 16.0, num_mfccs=num_mfccs)


def make_frontend(lib, win, hop, num_mfccs=40, method="mfcc"):
    return T.Frontend(window_size_samples=int(win), window_stride_samples=int(hop), num_mfccs=num_mfccs, method=method,
                      lib=lib, device=device_of(lib))


def fixture_params(fx, name, width):
    """(arch, params, stats) of a net fixture (regenerated from the seeds when not stored in full)."""
    arch = R.make_tcresnet(name, float(width))
    if bool(fx["full"]):
        p = {k[len("param:"):]: v for k, v in fx.items() if k.startswith("param:")}
        s = {k[len("stat:"):]: v for k, v in fx.items() if k.startswith("stat:")}
    else:
        p, s = R.init_params(arch, int(fx["init_seed"]))
        R.randomize_bn(arch, p, s, int(fx["bn_seed"]))
    return arch, p, s


def make_net(lib, name, width, t_in, p, s, in_channels=40, num_classes=12):
    net = T.TCResNet(name, R.tcresnet_channels(name, float(width)), in_channels, t_in, num_classes, lib=lib, device=device_of(lib))
    sd = dict(p)
    sd.update(s)
    net.load_state_dict(sd)
    return net


def check_frontend(lib, tag):
    fx = load(f"frontend_{tag}.npz")
    fe = make_frontend(lib, fx["win"], fx["hop"])
    feat = fe(to_dev(lib, fx["wav"]))
    got = fe.reference_view(feat)[..., 0].cpu().numpy()
    assert got.shape == fx["mfcc"].shape
    err = np.abs(got - fx["mfcc"]).max()
    assert err < MFCC_TOL, f"MFCC {tag}: max abs err {err}"
    # halo must be exactly zero (it implements SAME padding for conv0)
    f = feat.cpu().numpy()
    assert np.all(f[:, :, :T._lib.HALO] == 0) and np.all(f[:, :, T._lib.HALO + fe.n_frames:] == 0)
    # digital silence (row 2): every log-mel = ln(1e-6) -> c0 = -156.3, the rest 0 (SURVEY "hard parts")
    assert abs(got[2, 0, 0] - (-156.3047)) < 1e-2 and np.abs(got[2, :, 1:]).max() < 1e-3
    return err


def check_frontend_edges(lib, tag):
    """Low-amplitude / extreme-spectrum rows (oracle/make_golden.py::edge_waveforms).
    Rows 0-3 -- silent clips under background noise at volume 0.01 / 0.05 / 0.1 (the reference's `_silence_` workload) and 1e-4 noise --
    hold the same 1e-4 as the ordinary rows although log(mel + 1e-6) is steepest there; every kernel variant and the deploy path.
    Rows 4-5 are pure tones (a full-scale 1 kHz line; a 30 Hz line below the first filter): every other bin of the float64 spectrum is
    Hann leakage 100+ dB down, i.e. BELOW float32 round-off of the line (|X| ~ 320, eps |X| ~ 4e-5, power noise ~ 1e-9 against the 1e-6
    log offset), so any float32 FFT's answer is its own round-off there: a complete float32 NumPy pipeline with pocketfft's
    single-precision transform is 3-5e-4 from the float64 oracle on these rows (1.5e-5 on ordinary ones), the kernels 3-7e-4 -- the same
    class (tests/test_sensitivity.py::test_pure_tone_rows_what_float32_can_hold, test_emu_parity.py::test_pure_tone_rows_are_where_a_
    float32_fft_is).  Bound: 1e-3.  The deploy path has NO offset
    (log(max(x, 1e-12)), and TF's C++ ops compute in double): its empty bands are pure round-off there -- the float32 kernels were up
    to 0.5 off, so the deploy path runs in float64 since round 5 (frontend_deploy_f64_kernel: every row within 2e-5, i.e. float32
    output rounding); the float32 kernels stay behind TCR_TUNE_DEPLOY_F32 with the old bound."""
    fx = load(f"frontend_edge_{tag}.npz")
    wav = to_dev(lib, fx["wav"])
    errs = {}
    fe = make_frontend(lib, fx["win"], fx["hop"])
    got = fe.reference_view(fe(wav))[..., 0].cpu().numpy()
    errs["packed"] = np.abs(got - fx["mfcc"]).max(axis=(1, 2))
    try:
        lib.tcr_tune(1, 4)                                  # the general scalar-FP32 kernel
        g2 = fe.reference_view(fe(wav))[..., 0].cpu().numpy()
        errs["scalar"] = np.abs(g2 - fx["mfcc"]).max(axis=(1, 2))
    finally:
        lib.tcr_tune(1, 0)
    try:
        lib.tcr_tune(23, 1)                                 # the two-waves-per-SIMD packed kernel (frontend_pk.hip)
        g3 = fe.reference_view(fe(wav))[..., 0].cpu().numpy()
        errs["packed2"] = np.abs(g3 - fx["mfcc"]).max(axis=(1, 2))
    finally:
        lib.tcr_tune(23, 0)
    fd = make_frontend(lib, fx["win"], fx["hop"], method="mfcc_deploy")
    gd = fd.reference_view(fd(wav))[..., 0].cpu().numpy()
    errs["deploy"] = np.abs(gd - fx["mfcc_deploy"]).max(axis=(1, 2))            # the float64 kernel (default)
    try:
        lib.tcr_tune(26, 1)                                 # the float32 throughput kernels on the deploy path (rounds 3-4)
        g32 = fd.reference_view(fd(wav))[..., 0].cpu().numpy()
        errs["deploy_f32"] = np.abs(g32 - fx["mfcc_deploy"]).max(axis=(1, 2))
    finally:
        lib.tcr_tune(26, 0)
    for k, e in errs.items():
        assert e[:4].max() < MFCC_TOL, f"edge rows {tag} / {k}: per-row max abs err {e}"
        assert e[4:].max() < {"deploy": 2e-5, "deploy_f32": 0.5}.get(k, 1e-3), f"pure-tone rows {tag} / {k}: per-row max abs err {e}"
    assert errs["deploy"].max() < 2e-5                      # float64 inside, float32 only on the way out: every row
    return errs


def check_small_batch_kernel(lib, batches=(1, 5, 64)):
    """The small-batch network kernel (TCResNet8-1.0 at 49 frames, <= 64 utterances: one utterance per workgroup, weights DMA-copied
    into LDS a phase ahead, one tile per job) against the throughput kernel at one utterance per group (TCR_TUNE_NET_SMALL = 1) and
    against rows of a large batch: the same accumulation order, so logits, probabilities and ranges are bitwise equal."""
    arch = R.make_tcresnet("TCResNet8", 1.0)
    p, s = R.init_params(arch, 5)
    R.randomize_bn(arch, p, s)
    fe = make_frontend(lib, 640, 320)
    net = make_net(lib, "TCResNet8", 1.0, fe.n_frames, p, s)
    wav = to_dev(lib, R.synth_waveforms(max(max(batches), 80), seed=21))
    big = [t.clone() for t in net.forward_infer(fe(wav[:80].contiguous()), want_ranges=True)]
    for b in batches:
        feat = fe(wav[:b].contiguous())
        new = [t.clone() for t in net.forward_infer(feat, want_ranges=True)]
        try:
            lib.tcr_tune(27, 1)
            old = [t.clone() for t in net.forward_infer(feat, want_ranges=True)]
        finally:
            lib.tcr_tune(27, 0)
        for x, y, z, what in zip(new, old, big, ("logits", "probs", "ranges")):
            assert torch.equal(x, y), (b, what, float((x - y).abs().max()))
            assert torch.equal(x, z[:b]), (b, what, "vs rows of a batch of 80")


def check_edge_rows_logits(lib, tag):
    """The edge rows of the front-end fixtures (silence under background noise at three volumes, 1e-4 noise, two pure tones) through the
    WHOLE eval path -- front-end kernel -> TCResNet8-1.0 kernel with the net fixture's weights and randomised BN statistics -- against
    the oracle's logits on the oracle's float64 features: identical argmax on all six rows, logits within the ordinary tolerance on rows
    0-3; rows 4-5 (pure tones: features up to 1e-3 off in bands that hold float32 round-off, see check_frontend_edges) within 5e-4
    (measured 2e-4 / 2e-5)."""
    fx = load(f"frontend_edge_{tag}.npz")
    nfx = load(f"tcresnet8_1.0_{tag}.npz")
    arch, p, s = fixture_params(nfx, "TCResNet8", 1.0)
    ref = R.forward(arch, p, s, fx["mfcc"].astype(np.float64), is_training=False)["logits"]
    fe = make_frontend(lib, fx["win"], fx["hop"])
    net = make_net(lib, "TCResNet8", 1.0, fe.n_frames, p, s)
    got = net.forward_infer(fe(to_dev(lib, fx["wav"])))[0].cpu().numpy()
    err = np.abs(got - ref).max(axis=1)
    assert np.array_equal(got.argmax(1), ref.argmax(1)), (got.argmax(1), ref.argmax(1))
    assert err[:4].max() < LOGIT_TOL, f"edge rows {tag}: logits per-row max abs err {err}"
    assert err[4:].max() < 5e-4, f"pure-tone rows {tag}: logits per-row max abs err {err}"
    return err


def check_frontend_kernels_bitwise(lib, batch):
    """The three-waves-per-SIMD front-end (frontend_pk3.hip, default) against the two-waves kernel of rounds 2-4 (frontend_pk.hip,
    knob 23 = 1): the same operations in the same order on every value -- bitwise, for both reference framings, every method, the
    DS-CNN coefficient count and a ragged batch (partial last chunk, more chunks than workgroups at the larger size)."""
    wav = to_dev(lib, np.concatenate([R.synth_waveforms(min(batch, 64), seed=5)] * ((batch + 63) // 64), axis=0)[:batch].copy())
    cases = [(640, 320, "mfcc", 40), (480, 160, "mfcc", 40), (640, 320, "mfcc", 10), (640, 320, "log_mel_spectrogram", 40),
             (480, 160, "log_mel_spectrogram", 40), (640, 320, "mfcc_deploy", 40), (480, 160, "mfcc_deploy", 40), (320, 160, "mfcc", 40)]
    for win, hop, method, nc in cases:
        fe = make_frontend(lib, win, hop, method=method, num_mfccs=nc)
        try:
            lib.tcr_tune(26, 1)             # (deploy cases: the float32 kernels' filterbank / log-floor variant, not the float64 kernel)
            new = fe(wav).clone()
            lib.tcr_tune(23, 1)
            old = fe(wav).clone()
        finally:
            lib.tcr_tune(23, 0)
            lib.tcr_tune(26, 0)
        assert torch.equal(new, old), (win, hop, method, nc, float((new - old).abs().max()))


def check_frontend_deploy(lib, tag):
    """Deploy-path MFCC (audio_spectrogram + mfcc op semantics, method "mfcc_deploy") against oracle.mfcc_deploy."""
    fx = load(f"frontend_{tag}.npz")
    cfg = R.FRONTEND_4020 if tag == "4020" else R.FRONTEND_3010
    fe = make_frontend(lib, cfg.win, cfg.hop, method="mfcc_deploy")
    got = fe.reference_view(fe(to_dev(lib, fx["wav"])))[..., 0].cpu().numpy()
    ref = R.mfcc_deploy(fx["wav"], cfg)
    err = np.abs(got - ref).max()
    assert err < MFCC_TOL, f"deploy MFCC {tag}: max abs err {err}"
    # the op's filterbank, read back from the plan: every in-range bin's two weights sum to 1, bins outside are weightless
    m = fe.mel_matrix()
    rs = m.sum(axis=1)
    hz = 0.5 * cfg.sample_rate / (cfg.n_bins - 1)
    start, end = int(1.5 + cfg.lower_edge_hertz / hz), int(cfg.upper_edge_hertz / hz)
    assert np.all(rs[:start] == 0) and np.all(rs[end + 1:] == 0)
    inner = rs[start:end + 1]
    # (the bins below the first / above the last centre frequency feed one channel only)
    assert np.all(inner <= 1.0 + 1e-6) and np.mean(np.abs(inner - 1.0) < 1e-6) > 0.8 and np.all(inner > 0)
    return err


def check_eval(lib, fname, name, width):
    fx = load(fname)
    arch, p, s = fixture_params(fx, name, width)
    fe = make_frontend(lib, fx["win"], fx["hop"])
    feat = fe(to_dev(lib, fx["wav"]))
    net = make_net(lib, name, width, fe.n_frames, p, s)
    logits, probs, ranges = net.forward_infer(feat, want_ranges=True)
    logits, probs, ranges = logits.cpu().numpy(), probs.cpu().numpy(), ranges.cpu().numpy()
    err = np.abs(logits - fx["eval_logits"]).max()
    assert err < LOGIT_TOL, f"{fname}: eval logits max abs err {err}"
    assert np.array_equal(logits.argmax(1), fx["eval_logits"].argmax(1)), "argmax class ids differ"
    assert np.abs(probs - fx["eval_probs"]).max() < 1e-5
    assert np.abs(ranges - fx["eval_ranges"]).max() < 1e-5
    assert np.allclose(probs.sum(1), 1.0, atol=1e-5)
    return err


def check_train(lib, fname, name, width, steps=3, grad_rtol=2e-4):
    """fwd(train) + bwd + momentum for `steps` steps against the fixture (keep_prob 0.5, wd 1e-3, lr 0.1)."""
    fx = load(fname)
    arch, p, s = fixture_params(fx, name, width)
    fe = make_frontend(lib, fx["win"], fx["hop"])
    feat = fe(to_dev(lib, fx["wav"]))
    labels = to_dev(lib, fx["labels"])
    net = make_net(lib, name, width, fe.n_frames, p, s)
    keep, wd, lr, mu = float(fx["train_keep_prob"]), float(fx["train_weight_decay"]), float(fx["train_lr"]), float(fx["train_momentum"])
    seed, off = int(fx["train_seed"]), int(fx["train_sample_offset"])
    b = fx["wav"].shape[0]
    worst = 0.0
    for step in range(steps):
        logits, probs, loss_sum = net.forward_train(feat, labels, keep_prob=keep, seed=seed + step, sample_offset=off)
        net.backward()
        if step == 0:
            assert np.abs(logits.cpu().numpy() - fx["train_logits"]).max() < LOGIT_TOL
            assert abs(float(loss_sum) / b - float(fx["train_model_loss"])) < 1e-4
            assert abs(float(net.l2_loss(wd)) - float(fx["train_l2_loss"])) < 1e-5
            for k in [k for k in fx if k.startswith("grad:")]:
                n = k[len("grad:"):]
                ref = fx[k]
                got = net.grad_view(n).cpu().numpy().reshape(ref.shape).astype(np.float64)
                if R.is_l2_param(n):
                    got = got + wd * net._view(n).cpu().numpy().reshape(ref.shape)       # the fixture includes wd * w
                scale = max(np.abs(ref).max(), 1e-3)
                e = np.abs(got - ref).max() / scale
                worst = max(worst, e)
                assert e < grad_rtol, f"{n}: grad rel err {e}"
        net.sgd_momentum_step(lr, mu, wd)
        if step == 0:
            for k in [k for k in fx if k.startswith("stat1:")]:
                ref = fx[k]
                assert np.abs(net._view(k[len("stat1:"):]).cpu().numpy() - ref).max() < 1e-5 * max(1.0, np.abs(ref).max()), k
            for k in [k for k in fx if k.startswith("param1:")]:
                ref = fx[k]
                assert np.abs(net._view(k[len("param1:"):]).cpu().numpy().reshape(ref.shape) - ref).max() < 2e-5, k
    if steps == 3:
        for k in [k for k in fx if k.startswith("param3:")]:
            ref = fx[k]
            assert np.abs(net._view(k[len("param3:"):]).cpu().numpy().reshape(ref.shape) - ref).max() < 2e-4, k
        for k in [k for k in fx if k.startswith("stat3:")]:
            ref = fx[k]
            assert np.abs(net._view(k[len("stat3:"):]).cpu().numpy() - ref).max() < 1e-4 * max(1.0, np.abs(ref).max()), k
    return worst


def check_dscnn(lib, size):
    from oracle import dscnn_ref as D
    fx = load("dscnn_4020.npz")
    p, s = D.init_params(D.net_def(size), seed=int(fx["init_seed"]))
    fe = make_frontend(lib, fx["win"], fx["hop"], num_mfccs=10)
    feat = fe(to_dev(lib, fx["wav"]))
    net = T.DSCNN(size, fe.n_frames, 10, 12, lib=lib, device=device_of(lib))
    assert net.total_params() == int(fx[f"n_params_{size}"])
    sd = dict(p)
    sd.update(s)
    net.load_state_dict(sd)
    logits, probs = net.forward_infer(feat)
    err = np.abs(logits.cpu().numpy() - fx[f"logits_{size}"]).max()
    assert err < LOGIT_TOL, f"DSCNN-{size}: logits max abs err {err}"
    assert np.array_equal(logits.cpu().numpy().argmax(1), fx[f"logits_{size}"].argmax(1))
    assert np.abs(probs.cpu().numpy() - fx[f"probs_{size}"]).max() < 1e-5
    return err


def check_dscnn_train(lib, size, steps=3, grad_rtol=2e-4):
    """DS-CNN train-mode forward + backward + Adam for `steps` steps against the fixture (lr 5e-4)."""
    from oracle import dscnn_ref as D
    fx0, fx = load("dscnn_4020.npz"), load("dscnn_train_4020.npz")
    p, s = D.init_params(D.net_def(size), seed=int(fx["init_seed"]))
    pre = f"{size}:"
    wav = R.synth_waveforms(3, seed=int(fx[pre + "wav_seed"]))
    fe = make_frontend(lib, fx0["win"], fx0["hop"], num_mfccs=10)
    feat = fe(to_dev(lib, wav))
    labels = to_dev(lib, fx["labels"])
    net = T.DSCNN(size, fe.n_frames, 10, 12, lib=lib, device=device_of(lib))
    sd = dict(p)
    sd.update(s)
    net.load_state_dict(sd)
    lr = float(fx["train_lr"])
    b = wav.shape[0]
    keys = [k[len(pre):] for k in fx if k.startswith(pre)]
    worst = 0.0
    for step in range(steps):
        logits, probs, loss_sum = net.forward_train(feat, labels)
        net.backward()
        if step == 0:
            assert np.abs(logits.cpu().numpy() - fx[pre + "train_logits"]).max() < LOGIT_TOL
            assert abs(float(loss_sum) / b - float(fx[pre + "train_model_loss"])) < 1e-4
            for k in [k for k in keys if k.startswith("grad:")]:
                n = k[len("grad:"):]
                ref = fx[pre + k]
                got = net.grad_view(n).cpu().numpy().reshape(ref.shape).astype(np.float64)
                if n.endswith("/biases") and "fc1" not in n:
                    # bias ahead of a train-mode BN: zero gradient (round-off noise in TF); the library writes exactly 0
                    assert np.abs(ref).max() < 1e-10 and np.all(got == 0.0), n
                    continue
                scale = max(np.abs(ref).max(), 1e-3)
                e = np.abs(got - ref).max() / scale
                worst = max(worst, e)
                assert e < grad_rtol, f"{n}: grad rel err {e}"
        net.adam_step(lr, step + 1)
        if step == 0:
            for k in [k for k in keys if k.startswith("stat1:")]:
                ref = fx[pre + k]
                assert np.abs(net._view(k[len("stat1:"):]).cpu().numpy() - ref).max() < 1e-5 * max(1.0, np.abs(ref).max()), k
            for k in [k for k in keys if k.startswith("param1:")]:
                ref = fx[pre + k]
                name = k[len("param1:"):]
                d = np.abs(net._view(name).cpu().numpy().reshape(ref.shape) - ref)
                # Adam's first step moves every weight by lr * g / (|g| + 1e-8) = lr * sign(g): it is a function of the SIGN of
                # the gradient only, and the sign of an entry within f32 gradient error of zero (taken as 2e-5 of the tensor's
                # largest entry, ten times the error the gradient checks above measure) is not determined.  Those entries
                # may differ by up to 2 lr; ALL others must match to 2e-5.
                gk = pre + "grad:" + name
                if gk in fx and not (name.endswith("/biases") and "fc1" not in name):
                    g = np.abs(fx[gk])
                    sure = g > 2e-5 * max(g.max(), 1e-3)
                    assert d[sure].max(initial=0.0) < 2e-5, (k, d[sure].max())
                    assert d.max() < 2.5 * lr, (k, np.mean(~sure), d.max())
                else:           # (gradient not stored in the fixture: the same criterion without the per-entry attribution)
                    assert d.max() < 2.5 * lr and np.mean(d > 2e-5) < 0.03, (k, d.max(), np.mean(d > 2e-5))
    if steps == 3:
        for k in [k for k in keys if k.startswith("stat3:")]:
            ref = fx[pre + k]
            assert np.abs(net._view(k[len("stat3:"):]).cpu().numpy() - ref).max() < 1e-4 * max(1.0, np.abs(ref).max()), k
        for k in [k for k in keys if k.startswith("param3:")]:
            ref = fx[pre + k]
            d = np.abs(net._view(k[len("param3:"):]).cpu().numpy().reshape(ref.shape) - ref)
            assert d.max() < 4 * lr and np.mean(d > 1e-4) < 0.03, (k, d.max(), np.mean(d > 1e-4))
    return worst


def check_dscnn_train_live(lib, size, grad_rtol=2e-4):
    """DS-CNN train-mode forward + backward against the oracle evaluated on the spot (sizes without a committed fixture).
    The waveform seed is the one of 12 candidates whose BN pre-activations stay farthest from the ReLU kink."""
    import dataclasses
    from oracle import dscnn_ref as D
    blocks = D.net_def(size)
    cfg = dataclasses.replace(R.FRONTEND_4020, num_mfccs=10)
    labels = R.synth_labels(3).astype(np.float64)
    best = None
    for seed in range(2468, 2480):
        p, s = D.init_params(blocks, seed=0)
        wav = R.synth_waveforms(3, seed=seed)
        f = D.forward(blocks, p, s, R.mfcc(wav, cfg), True)
        margin = min(np.abs(c["xhat"] + p[k + "/beta"]).min() for k, c in f["cache"].items() if isinstance(c, dict))
        if best is None or margin > best[0]:
            best = (margin, wav, f, p, s)
    margin, wav, f, p, s = best
    assert margin > 2e-6, margin
    g = D.backward(blocks, p, f, labels)
    fe = make_frontend(lib, cfg.win, cfg.hop, num_mfccs=10)
    net = T.DSCNN(size, fe.n_frames, 10, 12, lib=lib, device=device_of(lib))
    sd = dict(p)
    sd.update(s)
    net.load_state_dict(sd)
    logits, probs, loss_sum = net.forward_train(fe(to_dev(lib, wav)), to_dev(lib, labels))
    net.backward()
    assert np.abs(logits.cpu().numpy() - f["logits"]).max() < LOGIT_TOL
    assert abs(float(loss_sum) / 3 - D.loss(f["logits"], labels)) < 1e-4
    worst = 0.0
    for n, ref in g.items():
        got = net.grad_view(n).cpu().numpy().reshape(ref.shape).astype(np.float64)
        if n.endswith("/biases") and "fc1" not in n:
            assert np.all(got == 0.0), n
            continue
        e = np.abs(got - ref).max() / max(np.abs(ref).max(), 1e-3)
        worst = max(worst, e)
        assert e < grad_rtol, f"{n}: grad rel err {e}"
    for k, ref in f["new_stats"].items():
        assert np.abs(net._view(k).cpu().numpy() - ref).max() < 1e-5 * max(1.0, np.abs(ref).max()), k
    return worst


# ---- live-oracle checks at sizes without a committed fixture --------------------------------------------------------------
def relu_margin(arch, fwd) -> float:
    """Distance of the closest ReLU input of a train-mode forward to the kink: an f32 implementation cannot be expected to
    land on the same side of an input within round-off of zero, and one flipped mask moves that channel's gradient."""
    cache = fwd["cache"]
    m = min(np.abs(cache[c.name]["z"]).min() for c in arch.convs() if c.bn and c.relu)
    return float(min(m, min(np.abs(cache[f"block{b.index}/out"]["pre"]).min() for b in arch.blocks)))


def relu_near(arch, fwd, tau=1e-5) -> int:
    """Number of ReLU inputs of a train-mode forward within `tau` of the kink."""
    cache = fwd["cache"]
    n = sum(int((np.abs(cache[c.name]["z"]) < tau).sum()) for c in arch.convs() if c.bn and c.relu)
    return n + sum(int((np.abs(cache[f"block{b.index}/out"]["pre"]) < tau).sum()) for b in arch.blocks)


def pick_waveforms(arch, p, s, cfg, batch, seeds=range(100, 112)):
    """The candidate waveform batch whose train-mode pre-activations stay farthest from the ReLU kink (as make_golden does)."""
    best = None
    for seed in seeds:
        wav = R.synth_waveforms(batch, seed=seed)
        f = R.forward(arch, p, s, R.mfcc(wav, cfg), True)
        mg = relu_margin(arch, f)
        if best is None or mg > best[0]:
            best = (mg, wav)
    return best


def check_small_batch(lib, batch, name="TCResNet8", width=1.0, tag="4020", seeds=range(100, 112)):
    """BASELINE.json configs[0] (batch = 1) and batch 2: eval forward + one training step (train-mode BN over `batch` x T'
    positions, backward, momentum) against the oracle evaluated on the spot."""
    cfg = R.FRONTEND_4020 if tag == "4020" else R.FRONTEND_3010
    arch = R.make_tcresnet(name, float(width))
    p, s = R.init_params(arch, 5)
    R.randomize_bn(arch, p, s, 6)
    margin, wav = pick_waveforms(arch, p, s, cfg, batch, seeds)
    assert margin > 1e-6, margin
    labels = R.synth_labels(batch).astype(np.float64)
    x = R.mfcc(wav, cfg)
    fe = make_frontend(lib, cfg.win, cfg.hop)
    feat = fe(to_dev(lib, wav))
    net = make_net(lib, name, width, fe.n_frames, p, s)
    ev = R.forward(arch, p, s, x, False)
    logits, probs = net.forward_infer(feat)
    assert np.abs(logits.cpu().numpy() - ev["logits"]).max() < LOGIT_TOL
    assert np.array_equal(logits.cpu().numpy().argmax(1), ev["logits"].argmax(1)) and np.abs(probs.cpu().numpy() - ev["probs"]).max() < 1e-5
    tr = R.forward(arch, p, s, x, True)
    tl, tp, loss_sum = net.forward_train(feat, to_dev(lib, labels), keep_prob=1.0)
    net.backward()
    assert np.abs(tl.cpu().numpy() - tr["logits"]).max() < LOGIT_TOL
    assert abs(float(loss_sum) / batch - R.loss(tr["logits"], labels, p, 0.0)[1]) < 1e-4
    g = R.backward(arch, p, tr, labels, 0.0)
    worst = 0.0
    for k, ref in g.items():
        got = net.grad_view(k).cpu().numpy().reshape(ref.shape).astype(np.float64)
        e = np.abs(got - ref).max() / max(np.abs(ref).max(), 1e-3)
        worst = max(worst, e)
        assert e < 2e-4, f"{k}: grad rel err {e} (batch {batch})"
    for k, ref in tr["new_stats"].items():
        assert np.abs(net._view(k).cpu().numpy() - ref).max() < 1e-5 * max(1.0, np.abs(ref).max()), k
    net.sgd_momentum_step(0.1, 0.9, 0.001)
    for k, ref in g.items():
        w1 = p[k] - 0.1 * (ref + (0.001 * p[k] if R.is_l2_param(k) else 0.0))
        assert np.abs(net._view(k).cpu().numpy().reshape(w1.shape) - w1).max() < 2e-5, k
    return worst


def check_staged_equals_unstaged(lib, name, width, batch, tag="4020", keep_prob=0.5, handoff="level", bwd_knob=None):
    """One replica: forward_train / backward run stage by stage through the sync-BN hand-off API with an identity hook must be
    BITWISE the unstaged path (logits, loss, every gradient, moving statistics)."""
    cfg = R.FRONTEND_4020 if tag == "4020" else R.FRONTEND_3010
    arch = R.make_tcresnet(name, float(width))
    p, s = R.init_params(arch, 2)
    R.randomize_bn(arch, p, s, 3)
    base = R.synth_waveforms(min(batch, 64), seed=21)
    reps = max(batch // base.shape[0], 1)
    wav = to_dev(lib, np.tile(base, (reps, 1)))
    labels = to_dev(lib, np.tile(R.synth_labels(base.shape[0]), (reps, 1)))
    fe = make_frontend(lib, cfg.win, cfg.hop)
    feat = fe(wav)
    outs = []
    seen = []
    # The per-unit stage API keeps the per-layer backward chain (reduce -> bn_bwd_apply -> data gradient); whole passes and dependency
    # levels run the lazy BN backward (bwd_lazy.hip).  Bitwise equality is a property of ONE chain, so the per-unit comparison pins the
    # unstaged run to the per-layer chain as well (TCR_TUNE_TRAIN_BWD = 2).
    lib.tcr_tune(9, bwd_knob if bwd_knob is not None else (2 if handoff == "unit" else 0))
    try:
        for hook in (None, lambda sums: seen.append((sums.dtype, sums.numel()))):
            net = make_net(lib, name, width, fe.n_frames, p, s)
            net.handoff = handoff
            logits, probs, loss = net.forward_train(feat, labels, keep_prob=keep_prob, seed=11, sync_hook=hook)
            g = net.backward().clone()
            outs.append((logits.clone(), probs.clone(), loss.clone(), g, net.stats.clone()))
    finally:
        lib.tcr_tune(9, 0)
    # one hand-off per dependency level each way: conv0, and per block (shortcut + first conv) | second conv
    nblocks = len(R.tcresnet_channels(name, float(width))) - 1
    assert lib.tcr_net_num_levels(net._h, 0) - 1 == 1 + 2 * nblocks == lib.tcr_net_num_levels(net._h, 1) - 1
    nbn = len([c for c in arch.convs() if c.bn])
    assert len(seen) == (2 * (1 + 2 * nblocks) if handoff == "level" else 2 * nbn) and all(dt == torch.float64 for dt, _ in seen)
    for a, b, what in zip(outs[0], outs[1], ("logits", "probs", "loss", "grads", "moving stats")):
        assert torch.equal(a, b), f"staged {what} differ from the unstaged run (max |d| {float((a - b).abs().max())})"
    assert torch.isfinite(outs[0][3]).all()


def check_dscnn_staged_equals_unstaged(lib, size, batch):
    """DS-CNN through the staged sync-BN API with an identity hook: bitwise the unstaged run (logits, loss, gradients, moving stats)."""
    from oracle import dscnn_ref as D
    p, s = D.init_params(D.net_def(size), seed=4)
    fe = make_frontend(lib, 640, 320, num_mfccs=10)
    base = R.synth_waveforms(min(batch, 32), seed=8)
    reps = -(-batch // base.shape[0])           # (ceil: the REQUESTED batch runs -- ragged sizes reach the kernels' fallback dispatch)
    feat = fe(to_dev(lib, np.tile(base, (reps, 1))[:batch]))
    labels = to_dev(lib, np.tile(R.synth_labels(base.shape[0]), (reps, 1))[:batch])
    assert feat.shape[0] == batch
    outs, seen = [], []
    for hook in (None, lambda sums: seen.append(sums.dtype)):
        net = T.DSCNN(size, fe.n_frames, 10, 12, lib=lib, device=device_of(lib))
        sd = dict(p); sd.update(s); net.load_state_dict(sd)
        logits, probs, loss = net.forward_train(feat, labels, sync_hook=hook)
        g = net.backward().clone()
        outs.append((logits.clone(), loss.clone(), g, net.stats.clone()))
    assert len(seen) == 2 * (net.lib.tcr_dscnn_num_stages(net._h) - 1) and all(dt == torch.float64 for dt in seen)
    for a, b, what in zip(outs[0], outs[1], ("logits", "loss", "grads", "moving stats")):
        assert torch.equal(a, b), f"DS-CNN-{size}: staged {what} differ from the unstaged run"


def check_dscnn_pointwise_wgrad_kernels(lib, size, batch, knob_id=25):
    """The DMA-staged pointwise filter gradient (TCR_TUNE_PW_WGRAD = 1: global_load_lds into two LDS buffers, three split-K wave groups)
    against the register-staged default: the same products, another summation order (split over wave groups and chunk counts) --
    every gradient tensor agrees to rounding, everything that does not pass through the kernel bitwise.
    knob_id 31 (TCR_TUNE_DW_WGRAD): the depthwise filter gradient by the row kernel (default) against the gather kernel, likewise."""
    from oracle import dscnn_ref as D
    p, s = D.init_params(D.net_def(size), seed=4)
    fe = make_frontend(lib, 640, 320, num_mfccs=10)
    base = R.synth_waveforms(min(batch, 32), seed=8)
    reps = -(-batch // base.shape[0])           # (ceil: the REQUESTED batch runs)
    feat = fe(to_dev(lib, np.tile(base, (reps, 1))[:batch]))
    labels = to_dev(lib, np.tile(R.synth_labels(base.shape[0]), (reps, 1))[:batch])
    assert feat.shape[0] == batch
    grads = []
    try:
        for knob in (0, 1):
            lib.tcr_tune(knob_id, knob)
            net = T.DSCNN(size, fe.n_frames, 10, 12, lib=lib, device=device_of(lib))
            sd = dict(p); sd.update(s); net.load_state_dict(sd)
            net.forward_train(feat, labels)
            grads.append(net.backward().clone())
    finally:
        lib.tcr_tune(knob_id, 0)
    a, b = grads
    assert torch.isfinite(a).all() and torch.isfinite(b).all()
    scale = float(a.abs().max())
    err = float((a - b).abs().max())
    assert scale > 0 and err <= 3e-5 * scale, (err, scale)
    assert not torch.equal(a, b)                # (the knob did select another kernel)


def check_dscnn_pointwise_geometries(lib, size, batch, knob_id=28, alt=1):
    """The wide pointwise conv kernel built for <= 128 registers (four waves per SIMD; default) against the unconstrained build of
    rounds 3-4 (TCR_TUNE_PW_POS = 1): the same instructions in another register allocation -- eval logits and training gradients bitwise.
    knob_id 29 (TCR_TUNE_BN_APPLY): the BN-backward apply pass with four float4 per thread and its coefficients in LDS (default) against
    the one-float4-per-thread kernel -- the same expression per element, gradients bitwise.  knob_id 30 (TCR_TUNE_DW_DGRAD): the depthwise
    data gradient of the stride-1 units by the row kernel (default) against the padded-image kernel -- same fmaf chain per output, same
    order of the backward sums, gradients bitwise.  knob_id 32 (TCR_TUNE_DW_FWD): the forward depthwise conv likewise (eval logits too).
    knob_id 25, alt 2 (TCR_TUNE_PW_WGRAD): the pointwise filter gradient's unrolled kernel (default) against its run-time-shape form."""
    from oracle import dscnn_ref as D
    p, s = D.init_params(D.net_def(size), seed=4)
    fe = make_frontend(lib, 640, 320, num_mfccs=10)
    base = R.synth_waveforms(min(batch, 32), seed=8)
    reps = max(batch // base.shape[0], 1)
    feat = fe(to_dev(lib, np.tile(base, (reps, 1))))
    labels = to_dev(lib, np.tile(R.synth_labels(base.shape[0]), (reps, 1)))
    evals, grads = [], []
    try:
        for knob in (0, alt):
            lib.tcr_tune(knob_id, knob)
            net = T.DSCNN(size, fe.n_frames, 10, 12, lib=lib, device=device_of(lib))
            sd = dict(p); sd.update(s); net.load_state_dict(sd)
            evals.append(net.forward_infer(feat)[0].clone())
            net.forward_train(feat, labels)
            grads.append(net.backward().clone())
    finally:
        lib.tcr_tune(knob_id, 0)
    assert torch.equal(evals[0], evals[1]), float((evals[0] - evals[1]).abs().max())
    assert torch.isfinite(grads[0]).all() and torch.equal(grads[0], grads[1]), float((grads[0] - grads[1]).abs().max())


def check_dscnn_lazy_equals_materialised(lib, size, batch, seed=6):
    """DS-CNN training with the normalised activations never materialised (default for the 172 / 276-channel nets: BN + ReLU applied by
    the consumers to the raw conv outputs, statistics / backward sums from the conv and data-gradient epilogues) against the materialising
    path (TCR_TUNE_DS_TRAIN = 1).  The arithmetic per element is the same; only the ORDER in which the per-channel sums are added
    differs, so everything agrees to rounding (gradients: a last-bit change of scale / shift may flip ReLU inputs within ~1e-7 of zero).
    tcr_dscnn_materialize_unit reproduces the activation the other path stores."""
    import tcresnet_amd as T
    dev = device_of(lib)
    rng = np.random.RandomState(seed)
    t, f = 49, 10
    feat = T.features_to_planar(torch.from_numpy(rng.uniform(-2, 2, (batch, t, f)).astype(np.float32)).to(dev), lib=lib)
    labels = torch.from_numpy(R.synth_labels(batch).astype(np.float32)).to(dev)
    outs = []
    try:
        for knob in (0, 1, 4):          # 4: the pointwise units' BN backward inside their data-gradient kernel
            lib.tcr_tune(15, knob)
            ds = T.DSCNN(size, t, f, 12, lib=lib, device=dev)
            ds.init_xavier(2)
            lg, _, loss = ds.forward_train(feat, labels)
            acts = [ds.unit_output(u, batch).clone() for u in (0, 1, 2, lib.tcr_dscnn_num_units(ds._h) - 1)]
            outs.append((lg.clone(), float(loss), ds.stats.clone(), ds.backward().clone(), acts))
    finally:
        lib.tcr_tune(15, 0)
    a, b, c4 = outs
    assert torch.equal(a[0], c4[0]) and float((a[3] - c4[3]).abs().max()) < 2e-5 * max(1.0, float(a[3].abs().max()))     # (same forward; dy rounded once either way)
    assert float((a[0] - b[0]).abs().max()) < 2e-5 and abs(a[1] - b[1]) < 1e-4 * max(1.0, abs(b[1]))
    assert float((a[2] - b[2]).abs().max()) < 2e-6 * max(1.0, float(b[2].abs().max()))
    assert float((a[3] - b[3]).abs().max()) < 5e-3 * max(1.0, float(b[3].abs().max()))
    gm = float(b[3].abs().max())
    assert float((a[3] - b[3]).abs().mean()) < 2e-5 * max(1.0, gm), (float((a[3] - b[3]).abs().mean()), gm)
    for x, y in zip(a[4], b[4]):
        assert x.shape == y.shape and float((x - y).abs().max()) < 1e-5 * max(1.0, float(y.abs().max()))
    return outs


def check_bn_backward_fused_equals_pair(lib, name, width, batch, seed=3, combos=((0, 0), (1, 0), (96, 0), (0, 1), (1, 1))):
    """BN backward with the finalize folded into the apply pass (default) is BITWISE the finalize + apply pair (TCR_TUNE_BWD_BN_FUSED
    = 1): same channel blocks, same slice order of the partial rows -- so also bitwise the staged (sync BN) path at one replica."""
    import tcresnet_amd as T
    dev = device_of(lib)
    rng = np.random.RandomState(seed)
    t, f = 25, 40
    x = torch.from_numpy(rng.uniform(-2, 2, (batch, t, f)).astype(np.float32)).to(dev)
    feat = T.features_to_planar(x, lib=lib)
    labels = torch.from_numpy(R.synth_labels(batch).astype(np.float32)).to(dev)
    ch = R.tcresnet_channels(name, float(width))
    grads = []
    try:
        for knob, mask in combos:
            lib.tcr_tune(11, knob)
            lib.tcr_tune(12, mask)          # 1: ReLU masks read back from the activations instead of recomputed from the raw outputs
            net = T.TCResNet(name, ch, f, t, 12, lib=lib, device=dev)
            net.init_xavier(1)
            net.forward_train(feat, labels, keep_prob=0.5, seed=9)
            grads.append(net.backward().clone())
    finally:
        lib.tcr_tune(11, 0)
        lib.tcr_tune(12, 0)
    for g in grads[1:]:
        assert torch.equal(grads[0], g), float((grads[0] - g).abs().max())


def check_down_dgrad_order(lib, name, width, batch, seed=4, t=25, variants=((2, 0), (1, 0), (0, 0), (2, 2))):
    """A block's shortcut conv writing the block-input gradient first (early, on the side stream; conv_a's data gradient adds onto
    it -- the default) is BITWISE conv_a first and the shortcut added behind it (TCR_TUNE_DOWN_DGRAD = 1): one addition, commuted."""
    import tcresnet_amd as T
    dev = device_of(lib)
    rng = np.random.RandomState(seed)
    f = 40
    x = torch.from_numpy(rng.uniform(-2, 2, (batch, t, f)).astype(np.float32)).to(dev)
    feat = T.features_to_planar(x, lib=lib)
    labels = torch.from_numpy(R.synth_labels(batch).astype(np.float32)).to(dev)
    ch = R.tcresnet_channels(name, float(width))
    grads = []
    try:
        for knob, streams in variants:     # 2: early for every width (0: only nets of <= 48 channels); streams 2: shortcut units on the second internal stream
            lib.tcr_tune(17, knob)
            lib.tcr_tune(7, streams)
            net = T.TCResNet(name, ch, f, t, 12, lib=lib, device=dev)
            net.init_xavier(1)
            net.forward_train(feat, labels, keep_prob=0.5, seed=9)
            grads.append(net.backward().clone())
    finally:
        lib.tcr_tune(17, 0)
        lib.tcr_tune(7, 0)
    for g in grads[1:]:
        assert torch.equal(grads[0], g), float((grads[0] - g).abs().max())
    assert float(grads[0].abs().max()) > 0


def check_dscnn_mask_paths_agree(lib, size, batch, seed=5):
    """DS-CNN backward with the ReLU masks recomputed from the raw BN inputs (default) is BITWISE the run that reads the activations."""
    import tcresnet_amd as T
    dev = device_of(lib)
    rng = np.random.RandomState(seed)
    t, f = 49, 10
    feat = T.features_to_planar(torch.from_numpy(rng.uniform(-2, 2, (batch, t, f)).astype(np.float32)).to(dev), lib=lib)
    labels = torch.from_numpy(R.synth_labels(batch).astype(np.float32)).to(dev)
    grads, stats, logits = [], [], []
    try:
        for mask in (0, 1, 2, 3, 4):        # 1: masks read back; 2: scalar elementwise BN kernels; 3: also the scalar reduction kernel;
                                            # 4: default + the 16-byte reduction kernel at any grid size (small batches: the emulator's)
            lib.tcr_tune(12, mask)
            ds = T.DSCNN(size, t, f, 12, lib=lib, device=dev)
            ds.init_xavier(2)
            lg, _, _ = ds.forward_train(feat, labels)
            logits.append(lg.clone()); stats.append(ds.stats.clone())
            grads.append(ds.backward().clone())
    finally:
        lib.tcr_tune(12, 0)
    assert torch.equal(grads[0], grads[1]), float((grads[0] - grads[1]).abs().max())
    assert torch.equal(grads[0], grads[2]), float((grads[0] - grads[2]).abs().max())
    # The scalar reduction kernel adds the same terms in another order.  The forward is continuous in the statistics: moving
    # statistics and logits agree to rounding.  The gradients go through ReLU masks taken from `fmaf(y, scale, shift) > 0`, so a last-bit
    # change of scale / shift flips inputs within ~1e-7 of zero (DS-CNN-L: millions of ReLU inputs): bounded, not bitwise.
    assert float((stats[4] - stats[3]).abs().max()) < 2e-6 * max(1.0, float(stats[3].abs().max()))
    assert float((logits[4] - logits[3]).abs().max()) < 2e-5
    assert float((grads[4] - grads[3]).abs().max()) < 5e-3 * max(1.0, float(grads[3].abs().max()))


def check_phase_kernel_variants(lib, name, width, batch, t=49, seed=8):
    """Training forward: the compile-time-shaped phase kernels with the bank-aligned utterance stride (default) are BITWISE the generic
    layer walk and the unpadded stride (TCR_TUNE_PHASE_STATIC bits 0 / 1): same jobs, same accumulation and statistics order; the
    phases' staging with float4 accesses + an LDS coefficient table (round 6; default) is bitwise the one-element loop (bit 2), alone
    and on top of the generic walk / the unpadded stride (the LDS image's 16-byte stores need a pitch that is a multiple of 4)."""
    import tcresnet_amd as T
    dev = device_of(lib)
    rng = np.random.RandomState(seed)
    f = 40
    x = torch.from_numpy(rng.uniform(-2, 2, (batch, t, f)).astype(np.float32)).to(dev)
    feat = T.features_to_planar(x, lib=lib)
    labels = torch.from_numpy(R.synth_labels(batch).astype(np.float32)).to(dev)
    ch = R.tcresnet_channels(name, float(width))
    outs = []
    try:
        for v in (0, 1, 2, 3, 4, 6):
            lib.tcr_tune(19, v)
            net = T.TCResNet(name, ch, f, t, 12, lib=lib, device=dev)
            net.init_xavier(1)
            logits, _, loss = net.forward_train(feat, labels, keep_prob=0.5, seed=9)
            outs.append((logits.clone(), float(loss), net.backward().clone(), net.stats.clone()))
    finally:
        lib.tcr_tune(19, 0)
    for o in outs[1:]:
        assert torch.equal(outs[0][0], o[0]) and outs[0][1] == o[1] and torch.equal(outs[0][2], o[2]) and torch.equal(outs[0][3], o[3])
    assert float(outs[0][2].abs().max()) > 0


def check_first_conv_wgrad_kernels_agree(lib, name, width, batch, t=49, seed=12):
    """The first conv's filter gradient from the LDS-staged nine-wave kernel (default) against the 16-byte-load kernel
    (TCR_TUNE_WGRAD_LDS = 1): another summation order, so to rounding -- and every other gradient bitwise.  Arm 4 runs the LDS-staged
    kernel with 16 utterances per workgroup (batch 4096's geometry: up to eight stages of two utterances, the 16-channel instance's
    two-stage lookahead in its steady state, a short last stage)."""
    import tcresnet_amd as T
    dev = device_of(lib)
    rng = np.random.RandomState(seed)
    f = 40
    x = torch.from_numpy(rng.uniform(-2, 2, (batch, t, f)).astype(np.float32)).to(dev)
    feat = T.features_to_planar(x, lib=lib)
    labels = torch.from_numpy(R.synth_labels(batch).astype(np.float32)).to(dev)
    ch = R.tcresnet_channels(name, float(width))
    grads = []
    try:
        for v in (0, 1, 4):
            lib.tcr_tune(21, v)
            net = T.TCResNet(name, ch, f, t, 12, lib=lib, device=dev)
            net.init_xavier(1)
            net.forward_train(feat, labels, keep_prob=0.5, seed=9)
            grads.append(net.backward().clone())
    finally:
        lib.tcr_tune(21, 0)
    n0 = 3 * f * ch[0]                      # the first conv's filter leads the parameter arena
    a, b, c = grads
    scale = float(b[:n0].abs().max())
    for g in (a, c):
        assert torch.equal(g[n0:], b[n0:])
        assert scale > 0 and float((g[:n0] - b[:n0]).abs().max()) <= 2e-5 * scale, (float((g[:n0] - b[:n0]).abs().max()), scale)
    assert not torch.equal(a[:n0], b[:n0]) or batch < 8


def check_backward_knob_variants(lib, knob, values, bitwise, name="TCResNet8", width=1.0, batch=19, t=49, seed=14, rtol=2e-5):
    """Backward variants behind one tcr_tune knob give the default's gradients: bitwise where the arithmetic order is the same
    (TCR_TUNE_LAZY_STAGE = 22), to rounding where the summation order differs (TCR_TUNE_WGRAD_WAVES = 20)."""
    import tcresnet_amd as T
    dev = device_of(lib)
    rng = np.random.RandomState(seed)
    f = 40
    x = torch.from_numpy(rng.uniform(-2, 2, (batch, t, f)).astype(np.float32)).to(dev)
    feat = T.features_to_planar(x, lib=lib)
    labels = torch.from_numpy(R.synth_labels(batch).astype(np.float32)).to(dev)
    ch = R.tcresnet_channels(name, float(width))
    grads = []
    try:
        for v in (0,) + tuple(values):
            lib.tcr_tune(knob, v)
            net = T.TCResNet(name, ch, f, t, 12, lib=lib, device=dev)
            net.init_xavier(1)
            net.forward_train(feat, labels, keep_prob=0.5, seed=9)
            grads.append(net.backward().clone())
    finally:
        lib.tcr_tune(knob, 0)
    scale = float(grads[0].abs().max())
    assert scale > 0
    for g in grads[1:]:
        if bitwise:
            assert torch.equal(grads[0], g), float((grads[0] - g).abs().max())
        else:
            assert float((grads[0] - g).abs().max()) <= rtol * scale, (float((grads[0] - g).abs().max()), scale)


def check_fused_eval_job_forms(lib, name, width, batch, t=49, seed=9):
    """Static fused eval kernels (fused.hip), the nine-tap layers' three job forms: work dealt in 16-position units + a whole tap of weight
    lookahead in the narrow layers (round-6 default), units without the lookahead (TCR_TUNE_NET_FUSED = 9), jobs of two tiles dealt
    round-robin (8: rounds 3-5).  The columns of the implicit GEMM are independent and every form accumulates tap-major, channel quads
    inner: BITWISE, for whole and ragged groups; the per-layer kernels (1) agree within the logit tolerance."""
    import tcresnet_amd as T
    dev = device_of(lib)
    rng = np.random.RandomState(seed)
    x = T.features_to_planar(torch.from_numpy(rng.uniform(-2, 2, (batch, t, 40)).astype(np.float32)).to(dev), lib=lib)
    net = T.TCResNet(name, R.tcresnet_channels(name, float(width)), 40, t, 12, lib=lib, device=dev)
    net.init_xavier(3)
    outs = {}
    try:
        for knob in (0, 8, 9, 1):
            lib.tcr_tune(3, knob)
            lg, pr = net.forward_infer(x)
            outs[knob] = (lg.clone(), pr.clone())
    finally:
        lib.tcr_tune(3, 0)
    for knob in (8, 9):
        assert torch.equal(outs[knob][0], outs[0][0]) and torch.equal(outs[knob][1], outs[0][1]), (name, batch, t, knob)
    assert float((outs[1][0] - outs[0][0]).abs().max()) < LOGIT_TOL, (name, batch, t)
