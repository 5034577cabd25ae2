"""The 2-D model families behind the factory -- ResNet2D8(/Pool), Res8/15(/Narrow), the KWSModel architectures --
on the generic graph engine, against the PyTorch float64 restatement (oracle/net2d_ref.py): eval logits / softmax / argmax,
train-mode logits / loss / every gradient / moving statistics, and the model classes through one optimisation step.
CPU (`-m "not gpu"`): the emulator build at reduced feature sizes; GPU (`-m gpu`): the gfx950 library at the reference's 98 x 40
(30 / 10 ms) and 49 x 40 shapes."""
import argparse

import numpy as np
import pytest
import torch

import tcresnet_amd as T
from oracle import net2d_ref as O
from oracle import numpy_ref as R
from tests import common as Cm


@pytest.fixture(params=["emu", pytest.param("hip", marks=pytest.mark.gpu)])
def rt(request):
    from tcresnet_amd import runtime
    from tcresnet_amd.audio_nets import tc_resnet
    lib = request.getfixturevalue("emu_lib" if request.param == "emu" else "hip_lib")
    runtime.set_default(lib, "cpu" if request.param == "emu" else "cuda")
    tc_resnet.reset_engines()
    yield lib
    runtime.set_default(None, None)
    tc_resnet.reset_engines()


def _randomise(eng, seed):
    """Non-trivial values for every variable (biases, BN gamma / beta / moving statistics included)."""
    rng = np.random.RandomState(seed)
    sd = {k: v for k, v in eng.state_dict().items() if k in eng.tensors}       # (not the constants of the reference graph, e.g. SVDF's runtime-memory)
    for k, v in sd.items():
        ti = eng.tensors[k]
        if ti.kind == 0:
            continue                                   # weights keep the reference's initializer
        if ti.kind in (1, 4):
            sd[k] = rng.uniform(0.5, 1.5, v.shape).astype(np.float32)
        elif k.endswith("bias") or k.endswith("biases"):
            sd[k] = rng.uniform(-0.05, 0.05, v.shape).astype(np.float32)
        else:
            sd[k] = rng.uniform(-0.5, 0.5, v.shape).astype(np.float32)
    eng.load_state_dict(sd)
    return sd


def _split(eng, sd):
    p = {k: v.astype(np.float64) for k, v in sd.items() if eng.tensors[k].arena == 0}
    s = {k: torch.tensor(v.astype(np.float64)) for k, v in sd.items() if eng.tensors[k].arena == 1}
    return p, s


def _check(lib, eng, oracle_fwd, t, f, batch, masks_of=None, grad_rtol=3e-4, seed=0):
    """oracle_fwd(params, stats, x, is_training, masks) -> dict."""
    dev = Cm.device_of(lib)
    sd = _randomise(eng, seed + 1)
    p, s = _split(eng, sd)
    rng = np.random.RandomState(seed)
    x = rng.uniform(-2.0, 2.0, (batch, t, f)).astype(np.float32)       # features (the front-end is covered elsewhere)
    planar = T.features_to_planar(torch.from_numpy(x).to(dev), lib=lib)
    xt = torch.tensor(x.astype(np.float64))
    tp = {k: torch.tensor(v) for k, v in p.items()}
    ev = oracle_fwd(tp, s, xt, False, None)
    logits, probs = eng.forward_infer(planar)
    err = np.abs(logits.cpu().numpy() - ev["logits"].numpy()).max()
    assert err < Cm.LOGIT_TOL, f"eval logits differ by {err}"
    assert np.abs(probs.cpu().numpy() - ev["probs"].numpy()).max() < 1e-5
    assert np.array_equal(logits.cpu().numpy().argmax(1), ev["logits"].numpy().argmax(1))
    # training: forward (batch statistics, dropout masks of the kernels' own generator), backward
    labels = R.synth_labels(batch, num_classes=eng.num_classes).astype(np.float64)
    tseed, off = 17, 5
    masks = masks_of(tseed, off, batch) if masks_of else None
    stats0 = eng.stats.clone()
    tl, tpb, loss_sum = eng.forward_train(planar, torch.from_numpy(labels.astype(np.float32)).to(dev), seed=tseed, sample_offset=off)
    # ReLU inputs within 1e-5 of zero: an f32 forward may fall on the other side of each of them, and one flipped mask moves a
    # weight-gradient entry by about one of its ~10^4 terms (~1 %; full-size planes have millions of ReLU inputs, tens of them
    # that close).  For exactly those elements the oracle takes the side the kernels took (their post-ReLU activations, read
    # through tcr_g2d_node_output); every other element keeps the oracle's own sign, so the tolerance stays tight at every size.
    kept = [(eng.node_output(n, batch, True) > 0).cpu() for n in eng.relu_nodes]
    g = eng.backward()
    O.KINK_LOG.update(on=True, near=0, total=0, decide=kept, idx=0, followed=0)
    out, model, _tot, grads = O.loss_and_grads(lambda pp: oracle_fwd(pp, s, xt, True, masks), p, labels)
    assert O.KINK_LOG["idx"] == len(kept), "the oracle and the graph disagree on the number of ReLUs"
    O.KINK_LOG.update(on=False, decide=None)
    near = O.KINK_LOG["near"]
    assert np.abs(tl.cpu().numpy() - out["logits"]).max() < Cm.LOGIT_TOL
    assert abs(float(loss_sum) / batch - model) < 1e-4
    worst = 0.0
    for k, ref in grads.items():
        got = eng.grad_view(k).cpu().numpy().reshape(ref.shape).astype(np.float64)
        e = np.abs(got - ref).max() / max(np.abs(ref).max(), 1e-3)
        worst = max(worst, e)
        assert e < grad_rtol, f"{k}: grad rel err {e}"
    for k, ref in out["new_stats"].items():
        assert np.abs(eng._view(k).cpu().numpy() - ref).max() < 1e-5 * max(1.0, np.abs(ref).max()), k
    # bitwise run-to-run reproducibility of the step (no float atomics)
    eng.stats.copy_(stats0)
    tl2, _, loss2 = eng.forward_train(planar, torch.from_numpy(labels.astype(np.float32)).to(dev), seed=tseed, sample_offset=off)
    g1 = g.clone()
    assert torch.equal(tl, tl2) and torch.equal(g1, eng.backward()) and float(loss_sum) == float(loss2)
    return worst


SIZES = {"emu": {"res": (12, 10, 2), "res2d": (12, 10, 2), "res2dpool": (40, 24, 3), "kws": (30, 20, 2)},
         "hip": {"res": (98, 40, 3), "res2d": (98, 40, 3), "res2dpool": (98, 40, 3), "kws": (49, 40, 4)}}


@pytest.mark.parametrize("variant", ["Res8", "Res15", "Res8Narrow", "Res15Narrow"])
def test_res(rt, variant):
    from tcresnet_amd.audio_nets import res
    if rt.kind == "emu" and variant in ("Res8Narrow", "Res15"):
        pytest.skip("the emulator runs Res8 (45 channels, pool) and Res15Narrow (19 channels, dilation); all four run on the GPU")
    layers, ch, pool, dil = res._VARIANTS[variant]
    # GPU: a 20 x 12 plane first, then the reference's 98 x 40 -- both to 3e-4
    for t, f, b in ([SIZES["emu"]["res"]] if rt.kind == "emu" else [(20, 12, 3), SIZES["hip"]["res"]]):
        eng = res.get_engine(variant, t, f, 12)
        assert eng.state_dict()["Res/f_conv/weights"].shape == (3, 3, 1, ch) and "Res/conv0_bn/gamma" not in eng.tensors
        assert "Res/conv1_bn/moving_mean" in eng.tensors and f"Res/conv{layers - 1}/weights" in eng.tensors
        _check(rt, eng, lambda p, s, x, tr, m: O.res_forward(p, s, x, variant, tr), t, f, b)


@pytest.mark.parametrize("pool", [False, True])
def test_resnet2d8(rt, pool):
    from tcresnet_amd.audio_nets import tc_resnet
    t, f, b = SIZES[rt.kind]["res2dpool" if pool else "res2d"]
    scope = "ResNet2D8Pool" if pool else "ResNet2D8"
    eng = tc_resnet.get_engine_2d(scope, t, f, 12, 1.0, 0.5, ([4, 4], 4) if pool else None)
    ch = O.resnet2d_channels(f, 1.0)
    assert eng.state_dict()[f"{scope}/conv0/weights"].shape == (3, 3, 1, ch[0]) and f"{scope}/block0/down/BatchNorm/gamma" in eng.tensors
    if f == 40:
        assert ch[0] == 23                               # int((3*40*16 + 10*16*24) / (9 + 10*24))
    masks_of = lambda seed, off, batch: [O.dropout_mask(seed, eng.dropout_nodes[0], off, batch, ch[-1], 0.5)]
    _check(rt, eng, lambda p, s, x, tr, m: O.resnet2d_forward(p, s, x, scope, 1.0, ((4, 4), 4) if pool else None, tr, 0.5, m), t, f, b,
           masks_of=masks_of)


@pytest.mark.parametrize("arch", ["conv", "trad_fpool3", "one_fstride4", "low_latency_conv", "tiny_conv", "single_fc", "low_latency_svdf"])
def test_kws(rt, arch):
    from tcresnet_amd.audio_nets import kws
    t, f, b = SIZES[rt.kind]["kws"]
    if rt.kind == "emu" and arch == "conv":
        t, f = 16, 10                    # (two SAME 64-channel convs: the heaviest graph for the emulator)
    ms = {"spectrogram_length": t, "fingerprint_width": f, "label_count": 12, "fingerprint_size": t * f}
    eng = kws.get_engine(ms, arch)
    nodes = eng.dropout_nodes

    def masks_of(seed, off, batch):
        return [O.dropout_mask(seed, n, off, batch, int(np.prod(eng.shape(n))), 0.5) for n in nodes]
    _check(rt, eng, lambda p, s, x, tr, m: O.kws_forward(p, x, arch, tr, m), t, f, b, masks_of=masks_of if nodes else None)
    if arch == "low_latency_svdf":          # the reference's variables, in their TF shapes (kws.py:577-583, 586-590, 604-607, 623-624)
        sd = eng.state_dict()
        assert sd["weights_frequency"].shape == (f, 2560) and sd["weights_time"].shape == (2560, t) and sd["bias"].shape == (1280,)
        assert sd["runtime-memory"].shape == (2560, 1, t) and not sd["runtime-memory"].any()
        assert sd["first_fc_weights"].shape == (1280, 256) and sd["final_fc_weights"].shape == (256, 12)


def test_model_classes_train_and_evaluate(rt, tmp_path):
    """Res8NarrowModel / KWSModel / ResNet2D8Model through the reference command line: build, 2 optimisation steps, checkpoint,
    evaluation (the reference's scripts train these with Adam / momentum; scripts/commands/Res8NarrowModel*.sh, KWSfstride4*.sh)."""
    from tcresnet_amd import evaluate_audio, train_audio
    from tcresnet_amd.audio_nets import tc_resnet
    from tcresnet_amd.common import tf_bundle
    from tests.test_boundary import REF_EVAL_CMD, REF_TRAIN_CMD
    small = "--window_size_ms 60 --window_stride_ms 60" if rt.kind == "emu" else "--window_size_ms 40 --window_stride_ms 20"
    mf = "--num_mfccs 12" if rt.kind == "emu" else "--num_mfccs 40"
    for model, flags, opt in (("Res8NarrowModel", "--weight_decay 0.00001", "--optimizer adam"),
                              ("KWSModel", "--architecture one_fstride4", "--optimizer adam"),
                              ("ResNet2D8Model", "--weight_decay 0.001 --width_multiplier 1.0", "--optimizer mom --momentum 0.9")):
        tc_resnet.reset_engines()
        d = tmp_path / model
        cmd = (REF_TRAIN_CMD.replace("--window_size_ms 40 --window_stride_ms 20", small).replace("--num_mfccs 40", mf)
               .replace("--optimizer mom --momentum 0.9", opt).replace("--max_step_from_restore 3", "--max_step_from_restore 2")
               .replace("--lr_list 0.1 0.01 0.001", "--lr_list 0.001 0.001 0.001").replace("--batch_size 6", "--batch_size 3")
               .replace("TCResNet8Model --weight_decay 0.001 --width_multiplier 1.0", f"{model} {flags}"))
        tr = train_audio.train(train_audio.parse_arguments(cmd.format(d=d).split()))
        assert tr.global_step == 2 and np.isfinite(float(tr.model.total_loss))
        ck = tf_bundle.read_checkpoint(str(d / f"{model}-2"))
        assert all(k in ck for k in tr.model.engine.tensors) and int(ck["global_step"]) == 2
        ecmd = (REF_EVAL_CMD.replace("--window_size_ms 40 --window_stride_ms 20", small).replace("--num_mfccs 40", mf)
                .replace("TCResNet8Model --weight_decay 0.001 --width_multiplier 1.0", f"{model} {flags}"))
        out = evaluate_audio.main(evaluate_audio.parse_arguments(ecmd.format(d=d).split()))
        assert out["step"] == 2 and out["num_evaluated"] == 6 and np.isfinite(out["total_loss"])


def test_kws_checkpoint_holds_matmul_weights_in_their_tf_shape(rt, tmp_path):
    """The reference declares the KWS fully-connected variables 2-D ([K, N] operands of tf.matmul, kws.py:79-81, 149-153, 260-262):
    a TF-written KWSModel checkpoint holds them that way, so the state dict, the optimiser slots and the restore path use the
    TF shape, not the kernels' [h, w, c, N] (the loader compared against the 4-D shape and rejected such a file)."""
    import logging
    import tcresnet_amd as T
    from tcresnet_amd.audio_nets import kws
    from tcresnet_amd.common import tf_bundle
    from tcresnet_amd.common.model_loader import Ckpt
    t, f = 20, 12
    ms = {"spectrogram_length": t, "fingerprint_width": f, "label_count": 12, "fingerprint_size": t * f, "sample_rate": 16000}
    for arch in ("single_fc", "low_latency_conv", "conv"):
        eng = kws.get_engine(ms, arch)
        _randomise(eng, 1)
        sd = eng.state_dict()
        fc = [n for n in sd if "fc_weights" in n or n in ("weights", "linear_weights", "first_linear_weights")]
        assert fc, arch
        for n in fc:
            assert sd[n].ndim == 2 and eng.tf_shape(n) == sd[n].shape, (arch, n, sd[n].shape)
        for n in sd:
            if n not in fc and "weights" in n:
                assert sd[n].ndim == 4, (arch, n)
        # a bundle written with the 2-D shapes (what tf.train.Saver writes) restores bit for bit ...
        prefix = str(tmp_path / arch / "KWSModel-5")
        tf_bundle.write_checkpoint(prefix, sd)
        other = eng
        other.init_variables(7); _randomise(other, 2)             # (other values; the restore must bring the saved ones back)
        assert not np.array_equal(other.state_dict()[fc[0]], sd[fc[0]])
        Ckpt(other, logger=logging.getLogger("t")).load(prefix)
        for n, v in other.state_dict().items():
            assert np.array_equal(v, sd[n]), (arch, n)
        # ... and one holding the 4-D kernel layout is rejected the way tf.train.Saver rejects a shape mismatch
        bad = dict(sd)
        ti = eng.tensors[fc[0]]
        bad[fc[0]] = sd[fc[0]].reshape([ti.shape[i] for i in range(ti.rank)])
        tf_bundle.write_checkpoint(prefix + "bad", bad)
        if bad[fc[0]].shape != sd[fc[0]].shape:
            with pytest.raises(ValueError, match="Total size"):
                Ckpt(other, logger=logging.getLogger("t")).load(prefix + "bad")


def test_graph_engine_cross_replica_bn_staged_equals_unstaged(rt):
    """Graph engine through the cross-replica BN hand-off API (tcr_g2d_*_stage) with an identity hook at one replica: BITWISE the
    unstaged run (logits, loss, every gradient, moving statistics); one hand-off per BN node each way, float64 sums."""
    from tcresnet_amd.audio_nets import res
    variant = "Res15Narrow" if rt.kind == "emu" else "Res8"
    t, f, b = (12, 10, 3) if rt.kind == "emu" else (98, 40, 8)
    eng = res.get_engine(variant, t, f, 12)
    sd = _randomise(eng, 4)
    dev = Cm.device_of(rt)
    x = np.random.RandomState(2).uniform(-2.0, 2.0, (b, t, f)).astype(np.float32)
    planar = T.features_to_planar(torch.from_numpy(x).to(dev), lib=rt)
    labels = torch.from_numpy(R.synth_labels(b).astype(np.float32)).to(dev)
    outs, seen = [], []
    for hook in (None, lambda sums: seen.append((sums.dtype, sums.numel()))):
        eng.load_state_dict(sd)
        logits, probs, loss = eng.forward_train(planar, labels, seed=5, sync_hook=hook)
        g = eng.backward().clone()
        outs.append((logits.clone(), probs.clone(), loss.clone(), g, eng.stats.clone()))
    nbn = rt.tcr_g2d_num_stages(eng._h) - 1
    assert nbn > 0 and len(seen) == 2 * nbn and all(dt == torch.float64 for dt, _ in seen)
    for a, c, what in zip(outs[0], outs[1], ("logits", "probs", "loss", "grads", "moving stats")):
        assert torch.equal(a, c), f"staged {what} differ from the unstaged run (max |d| {float((a - c).abs().max())})"
