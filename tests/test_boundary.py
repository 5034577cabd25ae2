"""The host-side drop-in layer (reference class / flag / file names): the reference's own command lines driven end to end --
train_audio.train -> TF-format checkpoints -> evaluate_audio (once / loop + BestKeeper) -- and the model classes against the
oracle.  Every test runs twice: on the CPU through the emulator build of the kernel sources (`-m "not gpu"`), and on a real
MI355X through the gfx950 library (`-m gpu`)."""
import argparse
import os

import numpy as np
import pytest
import torch

import tcresnet_amd as T
from oracle import numpy_ref as R
from tests import common as Cm

REF_TRAIN_CMD = ("--dataset_path synthetic --dataset_split_name train --output_name output/softmax --num_classes 12 "
                 "--train_dir {d} --num_silent 1854 --augmentation_method anchored_slice_or_pad_with_shift --preprocess_method mfcc "
                 "--num_mfccs 40 --clip_duration_ms 1000 --window_size_ms 40 --window_stride_ms 20 --batch_size 3 --boundaries 2 4 "
                 "--max_step_from_restore 3 --lr_list 0.1 0.01 0.001 --absolute_schedule --no-boundaries_epoch --max_to_keep 20 "
                 "--step_save_checkpoint 500 --step_evaluation 500 --optimizer mom --momentum 0.9 --step_save_summaries 1 "
                 "TCResNet8Model --weight_decay 0.001 --width_multiplier 1.0")
REF_EVAL_CMD = ("--dataset_path synthetic --dataset_split_name valid --output_name output/softmax --num_classes 12 --checkpoint_path {d} "
                "--num_silent 258 --augmentation_method anchored_slice_or_pad --preprocess_method mfcc --num_mfccs 40 --clip_duration_ms 1000 "
                "--window_size_ms 40 --window_stride_ms 20 --background_frequency 0.0 --background_max_volume 0.0 --max_step_from_restore 30000 "
                "--batch_size 3 --no-shuffle --valid_type once --evaluation_iterations 2 TCResNet8Model --weight_decay 0.001 --width_multiplier 1.0")


@pytest.fixture(params=["emu", pytest.param("hip", marks=pytest.mark.gpu)])
def rt(request):
    """The library the model classes run on: the emulator build on CPU, the gfx950 build on the GPU box."""
    from tcresnet_amd import runtime
    from tcresnet_amd.audio_nets import tc_resnet
    lib = request.getfixturevalue("emu_lib" if request.param == "emu" else "hip_lib")
    runtime.set_default(lib, "cpu" if request.param == "emu" else "cuda")
    tc_resnet.reset_engines()
    yield lib
    runtime.set_default(None, None)
    tc_resnet.reset_engines()


def _model_args(**kw):
    base = dict(sample_rate=16000, window_size_ms=40.0, window_stride_ms=20.0, preprocess_method="mfcc", num_mel_bins=64, num_mfccs=40,
                lower_edge_hertz=80.0, upper_edge_hertz=7600.0, output_name="output/softmax", num_classes=12, weight_decay=0.001,
                dropout_keep_prob=0.5, width_multiplier=1.0, label_smoothing=0.0, clip_duration_ms=1000, input_batch_size=1)
    base.update(kw)
    return argparse.Namespace(**base)


def test_available_nets_and_registry(rt):
    from tcresnet_amd.factory import audio_nets
    from tcresnet_amd.datasets import preprocessor_factory
    assert audio_nets._available_nets == ["KWSModel", "Res8Model", "Res8NarrowModel", "Res15Model", "Res15NarrowModel", "DSCNNSModel",
                                          "DSCNNMModel", "DSCNNLModel", "TCResNet8Model", "TCResNet14Model", "ResNet2D8Model", "ResNet2D8PoolModel"]
    assert all(hasattr(audio_nets, n) for n in audio_nets._available_nets)
    assert set(preprocessor_factory._available_preprocessors) == {"mfcc", "log_mel_spectrogram", "no_preprocessing"}
    with pytest.raises(NotImplementedError):
        preprocessor_factory.factory("spectrogram", "s", "n")


def test_reference_command_lines_train_then_evaluate(rt, tmp_path):
    from tcresnet_amd import train_audio, evaluate_audio
    from tcresnet_amd.common import tf_bundle
    args = train_audio.parse_arguments(REF_TRAIN_CMD.format(d=tmp_path).split())
    assert args.model == "TCResNet8Model" and args.weight_decay == 0.001 and args.lr_list == [0.1, 0.01, 0.001]
    trainer = train_audio.train(args)
    assert trainer.global_step == 3
    model = trainer.model
    assert (args.height, args.width, args.channels) == (49, 40, 1)          # args mutated like the reference (:83)
    # show_models counts every variable of the graph (65 264 trainable + 656 moving statistics)
    assert model.total_params == 65920 and model.engine.total_params() == 65264 and tuple(model.audio.shape) == (3, 49, 40, 1)
    assert np.isfinite(float(model.total_loss)) and float(model.total_loss) > float(model.model_loss)
    # TF-format checkpoint: <train_dir>/<ModelName>-<global_step>.{index,data-00000-of-00001} + the `checkpoint` state file
    assert sorted(p.name for p in tmp_path.iterdir()) == ["TCResNet8Model-3.data-00000-of-00001", "TCResNet8Model-3.index", "checkpoint"]
    assert tf_bundle.latest_checkpoint(str(tmp_path)) == str(tmp_path / "TCResNet8Model-3")
    saved = tf_bundle.read_checkpoint(str(tmp_path / "TCResNet8Model-3"))
    assert saved["TCResNet8/conv0/weights"].shape == (3, 1, 40, 16) and saved["TCResNet8/conv0/weights"].dtype == np.float32
    assert "TCResNet8/fc2/weights" in saved and int(saved["global_step"]) == 3 and saved["global_step"].dtype == np.int64
    assert saved["TCResNet8/block1/conv1_0/weights/Momentum"].shape == (9, 1, 24, 32)
    assert np.array_equal(saved["TCResNet8/conv0/BatchNorm/moving_mean"], model.engine.state_dict()["TCResNet8/conv0/BatchNorm/moving_mean"])
    # evaluation restores the checkpoint directory and reports the reference's metrics
    eargs = evaluate_audio.parse_arguments(REF_EVAL_CMD.format(d=tmp_path).split())
    out = evaluate_audio.main(eargs)
    assert out["num_evaluated"] == 6 and out["step"] == 3 and 0.0 <= out["accuracy"] <= 1.0 and np.isfinite(out["total_loss"])
    assert out["accuracy/valid"] == out["accuracy"] and 0.0 <= out["top5_accuracy/valid"] <= 1.0
    assert {f"mAP/valid/{a}" for a in ("macro", "micro", "weighted", "samples")} <= set(out) and "classification_report/valid" in out
    assert "precision/valid/yes" not in out         # 12 labels >= --maximum_num_labels_for_metric 10: no per-class ops
    # BestKeeper (save_best_keeper defaults to on): <ckpt_dir>/<split>/<metric>/{checkpoint files, checkpoint, scores.tsv}
    keep = tmp_path / "valid" / "accuracy" / "valid"
    assert sorted(p.name for p in keep.iterdir()) == ["TCResNet8Model-3.data-00000-of-00001", "TCResNet8Model-3.index", "checkpoint", "scores.tsv"]
    header, row = (keep / "scores.tsv").read_text().strip().split("\n")
    cols = dict(zip(header.split("\t"), row.split("\t")))
    assert header.split("\t") == sorted(cols) and cols["step"] == "3" and cols["model_size"] == "65920"
    assert abs(float(cols["accuracy/valid"]) - out["accuracy"]) < 1e-5
    # the reference's final test run evaluates the best-kept directory (scripts/commands/TCResNet8Model-1.0_*.sh, last line)
    targs = evaluate_audio.parse_arguments(REF_EVAL_CMD.replace("--dataset_split_name valid", "--dataset_split_name test").format(d=keep).split())
    tout = evaluate_audio.main(targs)
    assert tout["step"] == 3 and tout["accuracy/test"] == out["accuracy"] and (keep / "test" / "accuracy" / "test" / "scores.tsv").exists()


def test_evaluation_loop_watches_the_train_dir(rt, tmp_path):
    """`--valid_type loop`: every checkpoint the trainer leaves in the directory is evaluated once, in order; the loop ends with
    the checkpoint at --max_step_from_restore; BestKeeper keeps the better one per metric."""
    from tcresnet_amd import train_audio, evaluate_audio
    from tcresnet_amd.common import tf_bundle, tf_utils
    cmd = REF_TRAIN_CMD.replace("--step_save_checkpoint 500", "--step_save_checkpoint 1").replace("--max_step_from_restore 3", "--max_step_from_restore 2")
    trainer = train_audio.train(train_audio.parse_arguments(cmd.format(d=tmp_path).split()))
    latest, every = tf_bundle.read_checkpoint_state(str(tmp_path))
    assert latest == "TCResNet8Model-2" and every == ["TCResNet8Model-1", "TCResNet8Model-2"]
    ecmd = REF_EVAL_CMD.replace("--valid_type once", "--valid_type loop").replace("--max_step_from_restore 30000", "--max_step_from_restore 2")
    res = evaluate_audio.main(evaluate_audio.parse_arguments(ecmd.format(d=tmp_path).split()))
    assert [r["step"] for r in res] == [2]          # the watcher starts from the latest checkpoint; step 2 >= max_step ends the loop
    # a later checkpoint appears -> the iterator yields it; nothing new within the timeout -> it returns
    it = tf_utils.ckpt_iterator(tmp_path, timeout=0.2, seconds_to_sleep=0.05)
    assert next(it) == str(tmp_path / "TCResNet8Model-2")
    trainer.global_step = 5
    trainer.save()
    assert next(it) == str(tmp_path / "TCResNet8Model-5")
    assert list(it) == []
    # BestKeeper: an equal score is not "better" (epsilon), a worse loss is not kept, a better one replaces the old copy
    bk = tf_utils.BestKeeper({"accuracy/valid": "max", "metric_loss/valid/total_loss": "min"}, "valid", tmp_path)
    assert bk.metric_best["accuracy/valid"] == pytest.approx(res[0]["accuracy"], abs=1e-5)     # re-read from scores.tsv
    keep, which = bk.monitor("valid", {"accuracy/valid": res[0]["accuracy"], "metric_loss/valid/total_loss": res[0]["total_loss"] - 1.0})
    assert keep and which == {"accuracy/valid": False, "metric_loss/valid/total_loss": True}


def test_resume_equals_uninterrupted_run_adam(rt, tmp_path):
    """Save -> restore -> continue reproduces the uninterrupted run bit for bit: weights, Adam slots AND the step counter
    behind Adam's bias correction / the dropout stream come back (the reference restores beta1_power / beta2_power)."""
    from tcresnet_amd import train_audio
    from tcresnet_amd.audio_nets import tc_resnet
    base = REF_TRAIN_CMD.replace("--optimizer mom --momentum 0.9", "--optimizer adam --use_ema --ema_decay 0.9").replace("--lr_list 0.1 0.01 0.001", "--lr_list 0.01 0.005 0.001")
    full = train_audio.train(train_audio.parse_arguments(base.replace("--max_step_from_restore 3", "--max_step_from_restore 3").format(d=tmp_path / "a").split()))
    want = {k: v.copy() for k, v in full.model.engine.state_dict().items()}
    want_ema = full.model.engine.slots["ExponentialMovingAverage"].clone()
    tc_resnet.reset_engines()
    train_audio.train(train_audio.parse_arguments(base.replace("--max_step_from_restore 3", "--max_step_from_restore 2").format(d=tmp_path / "b").split()))
    tc_resnet.reset_engines()
    resume = base.replace("--max_step_from_restore 3", "--max_step_from_restore 1").replace("--optimizer adam", f"--checkpoint_path {tmp_path / 'b'} --optimizer adam")
    args = train_audio.parse_arguments(resume.format(d=tmp_path / "b").split())
    # (the synthetic input restarts with the process, as the reference's dataset does: replay the two consumed batches)
    from tcresnet_amd.datasets.synthetic import SyntheticAudioDataWrapper
    from tcresnet_amd.factory import audio_nets
    from tcresnet_amd.helper.trainer import SingleLabelAudioTrainer
    ds = SyntheticAudioDataWrapper(args, None, "train", True)
    wavs, labels = ds.get_input_and_output_op()
    model = audio_nets.TCResNet8Model(args, ds)
    model.build(wavs, labels, is_training=True)
    tr = SingleLabelAudioTrainer(model, None, args, ds, "train")
    assert tr.global_step == 2 and tr.global_step_from_checkpoint == 2 and model.step_count() == 2
    ds.next_batch(); ds.next_batch()
    tr.train()
    assert tr.global_step == 3
    got = model.engine.state_dict()
    assert all(np.array_equal(got[k], want[k]) for k in want), [k for k in want if not np.array_equal(got[k], want[k])][:3]
    assert torch.equal(model.engine.slots["ExponentialMovingAverage"], want_ema)
    # --use_ema on the evaluation side reads the shadow variables
    from tcresnet_amd.common.model_loader import Ckpt
    Ckpt(model.engine, use_ema=True, ema_decay=0.9).load(str(tmp_path / "b" / "TCResNet8Model-3"))
    k = "TCResNet8/conv0/weights"
    ti = model.engine.tensors[k]
    assert torch.equal(model.engine._view(k).flatten(), want_ema[ti.offset:ti.offset + ti.size])
    # scope filters / missing variables (common/model_loader.py:42-83,126-133)
    from tcresnet_amd.common import tf_bundle
    part = {k: v for k, v in want.items() if not k.startswith("TCResNet8/fc2")}
    tf_bundle.write_checkpoint(str(tmp_path / "p" / "part-7"), part)
    with pytest.raises(ValueError, match="missing variable"):
        Ckpt(model.engine).load(str(tmp_path / "p" / "part-7"))
    assert len(Ckpt(model.engine, exclude_scopes="TCResNet8/fc2").load(str(tmp_path / "p" / "part-7"))) == len(part)
    assert len(Ckpt(model.engine, ignore_missing_vars=True).load(str(tmp_path / "p" / "part-7"))) == len(part)
    assert Ckpt(model.engine, include_scopes="TCResNet8/conv0, TCResNet8/fc/").load(str(tmp_path / "p" / "part-7")) == [
        "TCResNet8/conv0/weights", "TCResNet8/fc/weights", "TCResNet8/conv0/BatchNorm/gamma", "TCResNet8/conv0/BatchNorm/beta",
        "TCResNet8/conv0/BatchNorm/moving_mean", "TCResNet8/conv0/BatchNorm/moving_variance"]


def test_trainer_flags_are_honoured_or_refused(rt, tmp_path):
    from tcresnet_amd import train_audio
    from tcresnet_amd.audio_nets import tc_resnet
    # rmsprop with its constructor overrides, in-training evaluation on the training graph every step, epoch stop rule
    cmd = (REF_TRAIN_CMD.replace("--optimizer mom --momentum 0.9", "--optimizer rmsprop --optimizer_decay 0.8 --optimizer_epsilon 0.001 --momentum 0.5")
           .replace("--step_evaluation 500", "--step_evaluation 1 --evaluation_iterations 2").replace("--lr_list 0.1 0.01 0.001", "--lr_list 0.001 0.001 0.001")
           .replace("--max_step_from_restore 3", "--max_step_from_restore 50 --max_epoch_from_restore 0.00025"))
    tr = train_audio.train(train_audio.parse_arguments(cmd.format(d=tmp_path / "r").split()))
    assert tr.global_step == 2                       # 2 steps x 3 utterances / 22246 samples = 0.00027 epochs >= 0.00025
    assert tr.last_eval["num_evaluated"] == 6 and "accuracy/train" in tr.last_eval and "metric_loss/train/total_loss" in tr.last_eval
    assert set(tr.model.engine.slots) >= {"RMSProp", "RMSProp_1"}
    tc_resnet.reset_engines()
    with pytest.raises(TypeError):                  # tf.train.AdamOptimizer(momentum=...) is a constructor error in the reference too
        train_audio.train(train_audio.parse_arguments(REF_TRAIN_CMD.replace("--optimizer mom", "--optimizer adam").format(d=tmp_path / "x").split()))
    # --trainable_scopes (common/tf_utils.py:19-37): only the variables re.match-ing a scope get an update / slot / decay step; the
    # others stay bit for bit (BN moving statistics are not variables of the optimiser: they keep updating)
    tc_resnet.reset_engines()
    tr = train_audio.train(train_audio.parse_arguments(REF_TRAIN_CMD.replace("--optimizer mom", "--trainable_scopes TCResNet8/fc,TCResNet8/block2/conv2_1/B --optimizer mom")
                                                       .replace("--max_step_from_restore 3", "--max_step_from_restore 2").format(d=tmp_path / "y").split()))
    eng = tr.model.engine
    fresh = type(eng)(eng.scope, eng.channels, eng.in_channels, eng.t_in, eng.num_classes, lib=eng.lib, device=eng.device)
    fresh.init_xavier(0)
    init, now, mom = fresh.state_dict(), eng.state_dict(), eng.slot_arena("Momentum")
    moved = {n for n in eng.trainable_names() if not np.array_equal(init[n], now[n])}
    assert moved == {"TCResNet8/fc/weights", "TCResNet8/fc2/weights", "TCResNet8/block2/conv2_1/BatchNorm/gamma", "TCResNet8/block2/conv2_1/BatchNorm/beta"} - \
        ({"TCResNet8/fc2/weights"} if np.array_equal(init["TCResNet8/fc2/weights"], now["TCResNet8/fc2/weights"]) and float(tr.args.weight_decay) == 0 else set())
    for n, ti in eng.tensors.items():
        if ti.arena == 0 and n not in moved:
            assert float(mom[ti.offset:ti.offset + ti.size].abs().max()) == 0.0, n      # no slot update either
    assert any(not np.array_equal(init[n], now[n]) for n in now if n.endswith("moving_mean"))
    # ... and the checkpoint holds slots for the trained variables only (tf.train.Optimizer creates slots for its var_list)
    from tcresnet_amd.common import tf_bundle
    ck = tf_bundle.read_checkpoint(str(tmp_path / "y" / "TCResNet8Model-2"))
    assert {k[:-len("/Momentum")] for k in ck if k.endswith("/Momentum")} == \
        {"TCResNet8/fc/weights", "TCResNet8/fc2/weights", "TCResNet8/block2/conv2_1/BatchNorm/gamma", "TCResNet8/block2/conv2_1/BatchNorm/beta"}
    # a scope that matches nothing: the reference's train op is tf.no_op() (helper/trainer.py:220-222) -- no variable, slot or moving
    # statistic changes; the losses are still fetched (here the step counter advances so that the run ends: documented deviation)
    tc_resnet.reset_engines()
    tr = train_audio.train(train_audio.parse_arguments(REF_TRAIN_CMD.replace("--optimizer mom", "--trainable_scopes NoSuchScope --optimizer mom")
                                                       .replace("--max_step_from_restore 3", "--max_step_from_restore 2").format(d=tmp_path / "z").split()))
    eng = tr.model.engine
    fresh = type(eng)(eng.scope, eng.channels, eng.in_channels, eng.t_in, eng.num_classes, lib=eng.lib, device=eng.device)
    fresh.init_xavier(0)
    init, now = fresh.state_dict(), eng.state_dict()
    assert tr.global_step == 2 and all(np.array_equal(init[n], now[n]) for n in now), [n for n in now if not np.array_equal(init[n], now[n])]
    with pytest.raises(SystemExit):
        train_audio.parse_arguments(REF_TRAIN_CMD.replace("--step_evaluation 500", "--step_evaluation 0").format(d=tmp_path).split())


def test_rmsprop_and_ema_kernels(rt):
    dev = Cm.device_of(rt)
    fe = Cm.make_frontend(rt, 640, 320)
    net = T.TCResNet("TCResNet8", [16, 24, 32, 48], 40, fe.n_frames, 12, lib=rt, device=dev)
    net.init_xavier(3)
    g = torch.Generator().manual_seed(0)
    net.grads.copy_(torch.randn(net.n_param, generator=g).to(dev) * 0.01)
    w0, gr = net.params.clone().cpu().double(), net.grads.clone().cpu().double()
    net.ema_init()
    for step in range(2):
        net.rmsprop_step(0.01, decay=0.8, momentum=0.5, eps=1e-3, weight_decay=0.001)
        net.ema_step(0.9)
    gg = gr.clone()
    w, ms, mom, sh = w0.clone(), torch.ones_like(w0), torch.zeros_like(w0), w0.clone()
    for step in range(2):
        gg = gr.clone()
        gg[:net.n_decay] += 0.001 * w[:net.n_decay]
        ms = 0.8 * ms + 0.2 * gg * gg
        mom = 0.5 * mom + 0.01 * gg / torch.sqrt(ms + 1e-3)
        w = w - mom
        sh = sh - 0.1 * (sh - w)
    assert (net.params.cpu().double() - w).abs().max() < 1e-6
    assert (net.slots["ExponentialMovingAverage"].cpu().double() - sh).abs().max() < 1e-6


def test_model_build_matches_oracle(rt):
    """TCResNet8Model(args).build(wavs, labels, is_training) -> logits/outputs/losses equal the oracle's."""
    from tcresnet_amd.factory import audio_nets
    dev = Cm.device_of(rt)
    fx = Cm.load("tcresnet8_1.0_4020.npz")
    arch, p, s = Cm.fixture_params(fx, "TCResNet8", 1.0)
    args = _model_args()
    model = audio_nets.TCResNet8Model(args)
    wavs = torch.from_numpy(fx["wav"]).unsqueeze(-1).to(dev)        # [B, 16000, 1] like the reference's dataset op
    labels = torch.from_numpy(fx["labels"].astype(np.float32)).to(dev)
    model.build(wavs, labels, is_training=False)            # creates the engine; then load the fixture weights
    sd = dict(p); sd.update(s)
    model.engine.load_state_dict(sd)
    model.build(wavs, labels, is_training=False)
    assert np.abs(model.logits.cpu().numpy() - fx["eval_logits"]).max() < Cm.LOGIT_TOL
    assert np.abs(model.outputs.cpu().numpy() - fx["eval_probs"]).max() < 1e-5
    assert np.array_equal(model.outputs.cpu().numpy().argmax(1), fx["eval_probs"].argmax(1))
    tot, mdl, l2 = R.loss(fx["eval_logits"], fx["labels"], p, 0.001)
    assert abs(float(model.model_loss) - mdl) < 1e-4 and abs(float(model.total_loss) - tot) < 1e-4
    assert tuple(model.inputs.shape) == (4, 49, 40, 1) and model.endpoints["ranges"].shape == (4, 2)
    # is_training=True build = the training graph (batch statistics, dropout keyed by the step counter), read-only
    stats0 = model.engine.stats.clone()
    model.build(wavs, labels, is_training=True)
    assert torch.equal(stats0, model.engine.stats)
    ref = R.forward(arch, p, s, fx["mfcc"], True)
    args1 = _model_args(dropout_keep_prob=1.0)
    m1 = audio_nets.TCResNet8Model(args1)
    m1.build(wavs, labels, is_training=True)
    assert m1.engine is model.engine
    assert np.abs(m1.logits.cpu().numpy() - ref["logits"]).max() < Cm.LOGIT_TOL
    assert abs(float(m1.model_loss) - R.loss(ref["logits"], fx["labels"], p, 0.0)[1]) < 1e-4
    args10 = argparse.Namespace(**{**vars(args), "num_mfccs": 10, "weight_decay": 0.0})
    ds = audio_nets.DSCNNLModel(args10)
    ds.build(wavs, labels, is_training=False)
    assert tuple(ds.audio.shape) == (4, 49, 10, 1) and ds.engine.total_params() == 413736 and tuple(ds.logits.shape) == (4, 12)
    # DS-CNN trains with Adam (scripts/commands/DSCNN*Model*.sh): the training build and one optimisation step run
    ds_s = audio_nets.DSCNNSModel(args10)
    ds_s.build(wavs, labels, is_training=True)
    before = ds_s.engine.params.clone()
    tot, mdl = ds_s.train_step(wavs, labels, 5e-4, optimizer="adam")
    assert np.isfinite(float(tot)) and abs(float(mdl) - float(ds_s.model_loss)) < 1e-6 and not torch.equal(before, ds_s.engine.params)
    # deploy-path MFCC through the reference's preprocessor call (for_deploy=True; the reference feeds batch 1 there)
    from tcresnet_amd.datasets import preprocessor_factory
    pre = preprocessor_factory.factory("mfcc", "input/audio/preprocessing", "input/audio/preprocessed")
    dep = pre.preprocess(wavs[:1], window_size_samples=640, window_stride_samples=320, for_deploy=True, **vars(args))
    assert tuple(dep.shape) == (1, 49, 40, 1)
    assert np.abs(dep[0, :, :, 0].cpu().numpy() - R.mfcc_deploy(fx["wav"][:1], R.FRONTEND_4020)[0]).max() < Cm.MFCC_TOL
    with pytest.raises(NotImplementedError):
        preprocessor_factory.factory("log_mel_spectrogram", "s", "n").preprocess(wavs[:1], 640, 320, True, **vars(args))


def test_deployable_model_and_freeze(rt, tmp_path):
    """build_deployable_model (both variants) + freeze.py: variables -> constants, `<checkpoint>.pb`, reloaded and run."""
    from tcresnet_amd import deploy, freeze, train_audio
    from tcresnet_amd.audio_nets import tc_resnet
    from tcresnet_amd.factory import audio_nets
    dev = Cm.device_of(rt)
    fx = Cm.load("tcresnet8_1.0_4020.npz")
    arch, p, s = Cm.fixture_params(fx, "TCResNet8", 1.0)
    # a checkpoint holding the fixture's variables, as a training run would leave it
    from tcresnet_amd.common import tf_bundle
    sd = {k: np.asarray(v, np.float32).reshape((v.shape[0], 1) + v.shape[1:]) if k.endswith("/weights") else np.asarray(v, np.float32) for k, v in {**p, **s}.items()}
    tf_bundle.write_checkpoint(str(tmp_path / "TCResNet8Model-30000"), sd)
    cmd = (f"--checkpoint_path {tmp_path / 'TCResNet8Model-30000'} --output_name output/softmax --num_classes 12 --preprocess_method no_preprocessing "
           "--height 49 --width 40 --channels 1 --window_size_ms 40 --window_stride_ms 20 --num_mfccs 40 TCResNet8Model --width_multiplier 1.0")
    out = freeze.freeze(freeze.parse_arguments(cmd.split()))
    assert out == str(tmp_path / "TCResNet8Model-30000.pb")
    tc_resnet.reset_engines()
    frozen = deploy.FrozenModel.load(out, lib=rt, device=dev)
    assert [(t.name, t.shape) for t in frozen.input_tensors] == [("input", (1, 49, 40, 1))] and frozen.output_tensor.name == "output/softmax"
    assert "TCResNet8/conv0/BatchNorm/gamma" not in frozen.constants and frozen.constants["TCResNet8/conv0/weights"].shape == (3, 1, 40, 16)
    feats = torch.from_numpy(fx["mfcc"].astype(np.float32)).unsqueeze(-1).to(dev)           # [B, 49, 40, 1]
    probs = frozen(feats).cpu().numpy()
    assert np.abs(probs - fx["eval_probs"]).max() < 1e-5 and np.array_equal(probs.argmax(1), fx["eval_probs"].argmax(1))
    # bitwise the variable-reading eval path
    net = Cm.make_net(rt, "TCResNet8", 1.0, 49, p, s)
    assert torch.equal(frozen(feats), net.forward_infer(T.features_to_planar(feats, lib=rt))[1])
    # include_preprocess=True: waveform placeholder [input_batch_size, 16000, 1] through the DEPLOY-path MFCC
    args = _model_args(input_batch_size=2)
    model = audio_nets.TCResNet8Model(args)
    inputs, output = model.build_deployable_model(include_preprocess=True)
    assert (inputs[0].name, inputs[0].shape) == ("input/audio/before_preprocessing", (2, 16000, 1)) and output.shape == (2, 12)
    model.engine.load_state_dict({**p, **s})
    fz = model.freeze()
    fz.save(str(tmp_path / "with_pre.pb"))
    fz2 = deploy.FrozenModel.load(str(tmp_path / "with_pre.pb"), lib=rt, device=dev)
    assert fz2.meta["frontend"]["method"] == "mfcc_deploy"
    wav = torch.from_numpy(fx["wav"][:2]).unsqueeze(-1).to(dev)
    ref = R.forward(arch, p, s, R.mfcc_deploy(fx["wav"][:2], R.FRONTEND_4020), False)
    assert np.abs(fz2(wav).cpu().numpy() - ref["probs"]).max() < 1e-5
    # DS-CNN freezes too
    args10 = _model_args(num_mfccs=10, weight_decay=0.0, height=49, width=10, channels=1)
    ds = audio_nets.DSCNNSModel(args10)
    ds.build_deployable_model(include_preprocess=False)
    f3 = ds.freeze()
    x = torch.randn(3, 49, 10, 1, device=dev)
    assert torch.equal(f3(x), ds.engine.forward_infer(T.features_to_planar(x, lib=rt))[1])
    # ... and the graph-engine families (rebuilt from the stored model name + arguments)
    kargs = _model_args(num_mfccs=12, height=16, width=12, channels=1, architecture="tiny_conv")
    km = audio_nets.KWSModel(kargs)
    km.build_deployable_model(include_preprocess=False)
    km.freeze().save(str(tmp_path / "kws.pb"))
    tc_resnet.reset_engines()
    kf = deploy.FrozenModel.load(str(tmp_path / "kws.pb"), lib=rt, device=dev)
    xk = torch.randn(2, 16, 12, 1, device=dev)
    assert torch.equal(kf(xk), km.engine.forward_infer(T.features_to_planar(xk, lib=rt))[1])


def test_lr_schedule():
    from tcresnet_amd.helper.trainer import piecewise_constant
    assert [piecewise_constant(s, [10000, 20000], [0.1, 0.01, 0.001]) for s in (0, 10000, 10001, 20000, 20001)] == [0.1, 0.1, 0.01, 0.01, 0.001]


def test_audio_metrics_match_reference_formulas():
    """Keys and values of the reference's non-tensor metric ops (metrics/ops/non_tensor_ops.py, metrics/funcs.py)."""
    from tcresnet_amd.metrics import audio_metrics, best_keep_metric_with_modes, top_n_accuracy
    rng = np.random.RandomState(0)
    names = ["__null__", "yes", "no", "up", "down", "left"]
    labels = rng.randint(0, 6, 64)
    onehot = np.eye(6)[labels]
    logits = rng.randn(64, 6) + 2.0 * onehot
    probs = np.exp(logits) / np.exp(logits).sum(1, keepdims=True)
    m = audio_metrics(probs, onehot, names, "valid")
    pred = probs.argmax(1)
    assert m["accuracy/valid"] == np.mean(pred == labels)
    assert m["top5_accuracy/valid"] == np.mean([labels[i] in np.argsort(-probs[i])[:5] for i in range(64)]) == top_n_accuracy(labels, probs, 5)
    tp = np.sum((pred == 1) & (labels == 1))
    assert abs(m["precision/valid/yes"] - tp / max(np.sum(pred == 1), 1)) < 1e-12 and abs(m["recall/valid/yes"] - tp / np.sum(labels == 1)) < 1e-12
    order = np.argsort(-probs[:, 2])                            # average precision of class "no" by its definition
    hits = (labels[order] == 2)
    ap = np.sum(np.cumsum(hits)[hits] / (np.nonzero(hits)[0] + 1)) / hits.sum()
    assert abs(m["ap/valid/no"] - ap) < 1e-12
    assert abs(m["mAP/valid/macro"] - np.mean([m[f"ap/valid/{n}"] for n in names])) < 1e-12
    assert m["classification_report/valid"].startswith("[ClassificationReport]")
    modes = best_keep_metric_with_modes("valid", names, True)
    assert modes["accuracy/valid"] == "max" and modes["metric_loss/valid/total_loss"] == "min" and modes["f1score/valid/up"] == "max"
    assert set(modes) - {"metric_loss/valid/total_loss", "metric_loss/valid/model_loss"} <= set(m)


def _write_wav(path, pcm, rate=16000):
    import struct
    pcm = np.asarray(pcm, dtype="<i2")
    body = b"WAVE" + b"fmt " + struct.pack("<IHHIIHH", 16, 1, 1, rate, rate * 2, 2, 16) + b"data" + struct.pack("<I", pcm.nbytes) + pcm.tobytes()
    path.write_bytes(b"RIFF" + struct.pack("<I", len(body)) + body)


def test_wav_directory_dataset_train_and_evaluate(rt, tmp_path):
    """The reference's on-disk layout (<path>/<split>/<label>/*.wav, _background_noise_, silent samples) through the
    reference command lines: decode + crop/pad + shift + background mix on the device input stage, then train / evaluate."""
    from tcresnet_amd import train_audio, evaluate_audio
    from tcresnet_amd.datasets.audio_data_wrapper import SingleLabelAudioDataWrapper
    rng = np.random.RandomState(0)
    words = ["down", "go", "left", "no", "off", "on", "right", "stop", "unknown", "up", "yes"]      # + __null__ = 12 classes
    for split, per in (("train", 2), ("valid", 1)):
        for w in words:
            d = tmp_path / "data" / split / w
            d.mkdir(parents=True)
            for k in range(per):
                _write_wav(d / f"{k}.wav", rng.randint(-8000, 8000, rng.randint(12000, 18000)))
        bg = tmp_path / "data" / split / "_background_noise_"
        bg.mkdir()
        _write_wav(bg / "noise.wav", rng.randint(-3000, 3000, 40000))
        _write_wav(bg / "short.wav", rng.randint(-3000, 3000, 100))             # shorter than a clip: cannot be cropped, skipped
    cmd = REF_TRAIN_CMD.replace("--dataset_path synthetic", f"--dataset_path {tmp_path / 'data'}").replace("--num_silent 1854", "--num_silent 2")
    args = train_audio.parse_arguments(cmd.format(d=tmp_path / "ckpt").split())
    ds = SingleLabelAudioDataWrapper(args, None, "train", True)
    assert ds.label_names == ["__null__"] + words and ds.num_samples == 24 and len(ds.background) == 1
    peek = ds.get_input_and_output_op()             # building the graph does not consume the dataset
    wavs, labels = ds.next_batch()
    assert torch.equal(peek[0], wavs) and torch.equal(peek[1], labels)
    assert tuple(wavs.shape) == (3, 16000, 1) and tuple(labels.shape) == (3, 12) and float(wavs.abs().max()) <= 1.0
    assert torch.all(labels.sum(1) == 1)
    # data-parallel shards: one global order, different augmentation draws per rank
    ds.setup_iterator()
    w0, l0 = ds.next_batch(0, 2)
    ds.setup_iterator()
    w1, l1 = ds.next_batch(1, 2)
    assert not torch.equal(l0, l1) or not torch.equal(w0, w1)
    ds.setup_iterator()
    assert torch.equal(ds.next_batch()[0], wavs)
    # a silent sample without background is all zeros; with the mix it is the scaled background crop
    sil = [i for i, f in enumerate(ds.filenames) if f == ""]
    from tcresnet_amd.datasets import augmentation_factory as F
    z = F.anchored_slice_or_pad(ds.pool, sil[:1], 16000, background_data=ds.background, is_training=False)
    assert float(z.abs().max()) == 0.0
    trainer = train_audio.train(args)
    assert trainer.global_step == 3 and np.isfinite(float(trainer.model.total_loss))
    ecmd = REF_EVAL_CMD.replace("--dataset_path synthetic", f"--dataset_path {tmp_path / 'data'}").replace("--num_silent 258", "--num_silent 1")
    out = evaluate_audio.main(evaluate_audio.parse_arguments(ecmd.format(d=tmp_path / "ckpt").split()))
    assert out["num_evaluated"] == 6 and "accuracy/valid" in out and np.isfinite(out["total_loss"])
    with pytest.raises(ValueError):
        bad = train_audio.parse_arguments(cmd.replace("--num_classes 12", "--num_classes 10").format(d=tmp_path / "c2").split())
        SingleLabelAudioDataWrapper(bad, None, "train", True)
