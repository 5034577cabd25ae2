"""TEST INFRASTRUCTURE ONLY -- the recipe that would PIN the oracle: regenerate tests/golden/*.npz from the REAL reference.

CANNOT RUN IN THIS IMAGE: it imports /root/reference (hyperconnect/TC-ResNet) under TensorFlow 1.13.1 / Python <= 3.7
(requirements/py36-cpu.txt:2); neither is installable here (no network, Python 3.10).  It must never travel to the GPU box
(nothing under oracle/ is imported by the product path; this file additionally refuses to run without TF 1.x).  Run it on a box
that has the reference's environment:

    cd /path/to/TC-ResNet && PYTHONPATH=/path/to/repo:. python /path/to/repo/oracle/pin_from_reference.py --out /path/to/repo/tests/golden

What it does, with the SAME seeds and inputs as oracle/make_golden.py (so every fixture keeps its keys and shapes):
  * front-end: feeds `wav` of frontend_{3010,4020}.npz and frontend_edge_*.npz through
    datasets.preprocessor_factory.factory("mfcc", ...).preprocess(...) (datasets/preprocessors.py:183-194; for_deploy=False and
    True) in a tf.Session on CPU and stores the fetched MFCCs under the keys `mfcc` / `mfcc_deploy` / `log_mel_magnitude`;
  * nets: builds audio_nets.tc_resnet.TCResNet8 / TCResNet14 (audio_nets/tc_resnet.py:57-70) under TCResNet_arg_scope
    (:102-123) on a placeholder, assigns the fixture's variables by TF name (the oracle's parameter dict uses the same
    names), and fetches eval-mode logits / softmax, train-mode logits, tf.gradients of the total loss
    (factory/audio_nets.py:147-183: softmax_cross_entropy + weight_decay * l2_loss), the variables after 1 and 3 MomentumOptimizer
    steps and the BN moving statistics (slim update ops) -- keys eval_logits, eval_probs, train_logits, grad:*, param1:*, param3:*,
    stat1:*, stat3:*.  Dropout: the fixtures use keep_prob 0.5 with the kernel's counter-based mask; TF cannot reproduce that stream,
    so the pinned train-mode vectors are generated with keep_prob 1.0
    (`pin_nets`: keys `tf:*_keep1`; the keep_prob 0.5 fixtures keep exercising the kernel's own mask against the restatement);
  * DS-CNN (`pin_dscnn`): builds audio_nets.ds_cnn.DSCNN with S / M / L_NET_DEF under DSCNN_arg_scope (audio_nets/ds_cnn.py:36-118,
    factory/audio_nets.py:299-360) on the 49 x 10 MFCC of dscnn_4020.npz, assigns the seed-generated variables by TF name and fetches the
    eval logits / softmax (`tf:logits_{S,M,L}`, `tf:probs_*`); for size S also the training graph of dscnn_train_4020.npz: train-mode
    logits, tf.gradients of the loss wrt every variable and the variables / moving statistics after one and three AdamOptimizer steps
    (lr 5e-4, TF defaults beta1 .9, beta2 .999, epsilon 1e-8: the reference's DS-CNN scripts) -- `tf:S:train_logits`, `tf:S:grad:*`,
    `tf:S:param{1,3}:*`, `tf:S:stat{1,3}:*`;
  * deploy-path MFCC (`for_deploy=True`: contrib_audio.audio_spectrogram + mfcc, datasets/preprocessors.py:98-124,196-203): pinned by
    `pin_frontend` as `mfcc_deploy` for every front-end fixture that carries the key (the edge-row fixtures included).
After it ran, `python -m pytest tests/test_oracle.py` checks the NumPy restatement against the now-pinned vectors: every
difference > 1e-10 (float64 graph) / > 1e-5 (float32 graph) is a place where SURVEY App. A's recollection of TF 1.13 semantics is wrong
(the four flagged ones: Hann / mel-matrix precision, SAME padding side, moving-variance estimator, dropout arithmetic;
tests/test_sensitivity.py says how far each can move the outputs).
"""
from __future__ import annotations

import argparse
import os
import sys

import numpy as np


def _require_reference():
    try:
        import tensorflow as tf
    except ImportError as e:                                 # the normal outcome in this repository's image
        raise SystemExit("pin_from_reference.py needs the reference's environment (tensorflow==1.13.1, Python <= 3.7): " + str(e))
    if not tf.__version__.startswith("1."):
        raise SystemExit(f"pin_from_reference.py needs TensorFlow 1.13-1.15 (tf.contrib), found {tf.__version__}")
    try:
        from audio_nets import ds_cnn, tc_resnet                           # noqa: F401  (the reference's modules: cwd = its checkout)
        from datasets import preprocessor_factory                          # noqa: F401
    except ImportError as e:
        raise SystemExit("run from the root of a hyperconnect/TC-ResNet checkout (audio_nets/, datasets/ importable): " + str(e))
    return tf


def pin_frontend(tf, golden: str, out: str):
    from datasets import preprocessor_factory
    for name in sorted(os.listdir(golden)):
        if not name.startswith("frontend_"):
            continue
        fx = dict(np.load(os.path.join(golden, name)))
        win, hop = int(fx["win"]), int(fx["hop"])
        res = {}
        for key, method, deploy, kw in (("mfcc", "mfcc", False, {}), ("mfcc_deploy", "mfcc", True, {}),
                                        ("log_mel_magnitude", "log_mel_spectrogram", False, {})):
            if key not in fx:
                continue
            tf.reset_default_graph()
            wav = tf.placeholder(tf.float32, [None, 16000, 1])
            pre = preprocessor_factory.factory(method, scope="pin", preprocessed_node_name="pre")
            node = pre.preprocess(wav, win, hop, for_deploy=deploy, num_mel_bins=64, sample_rate=16000, lower_edge_hertz=80.0,
                                  upper_edge_hertz=7600.0, num_mfccs=fx[key].shape[-1], **kw)
            with tf.Session(config=tf.ConfigProto(device_count={"GPU": 0})) as sess:
                res[key] = sess.run(node, {wav: fx["wav"][..., None]})[..., 0].astype(np.float64)
        fx.update(res)
        fx["tf:pinned"] = np.asarray(sorted(res), dtype="U")          # which keys now hold the reference's own outputs
        np.savez_compressed(os.path.join(out, name), **fx)
        print("pinned", name, {k: v.shape for k, v in res.items()})


NETS = (("tcresnet8_1.0_4020.npz", "TCResNet8", 1.0), ("tcresnet8_1.0_3010.npz", "TCResNet8", 1.0), ("tcresnet14_1.5_4020.npz", "TCResNet14", 1.5))


def pin_nets(tf, golden: str, out: str):
    """Adds `tf:*` keys to the net fixtures: eval logits / softmax / sigmoid ranges, and -- with dropout switched off (keep_prob 1.0:
    TF's dropout stream cannot be made to draw the kernel's counter-based mask) -- train-mode logits, losses, d(total_loss)/d(variable),
    variables and BN moving statistics after 1 and 3 MomentumOptimizer steps (lr 0.1, momentum 0.9, weight decay 0.001; the reference's
    train op, helper/trainer.py:199-222).  tests/test_oracle.py::test_oracle_against_pinned_reference compares the restatement with
    these keys whenever a fixture holds them."""
    slim = tf.contrib.slim
    from audio_nets import tc_resnet
    from oracle import numpy_ref as R
    for fname, name, width in NETS:
        fx = dict(np.load(os.path.join(golden, fname)))
        arch = R.make_tcresnet(name, width)
        if bool(fx["full"]):
            p = {k[len("param:"):]: v for k, v in fx.items() if k.startswith("param:")}
            st = {k[len("stat:"):]: v for k, v in fx.items() if k.startswith("stat:")}
        else:
            p, st = R.init_params(arch, int(fx["init_seed"]))
            R.randomize_bn(arch, p, st, int(fx["bn_seed"]))
        x = fx["mfcc"].astype(np.float32)[..., None]                         # [B, T, 40, 1] = model.audio (factory/audio_nets.py:49-60)
        labels = fx["labels"].astype(np.float32)
        wd, lr, mu = float(fx["train_weight_decay"]), float(fx["train_lr"]), float(fx["train_momentum"])
        res = {}
        for is_training in (False, True):
            tf.reset_default_graph()
            inp = tf.placeholder(tf.float32, [None] + list(x.shape[1:]))
            lab = tf.placeholder(tf.float32, [None, labels.shape[1]])
            with slim.arg_scope(tc_resnet.TCResNet_arg_scope(is_training, weight_decay=wd, keep_prob=1.0)):
                logits, endpoints = getattr(tc_resnet, name)(inp, labels.shape[1], width_multiplier=width)
            probs = slim.softmax(logits)
            model_loss = tf.losses.softmax_cross_entropy(logits=logits, onehot_labels=lab, label_smoothing=0.0, weights=1.0)
            not_bn = lambda n: ("batch_normalization" not in n) and ("BatchNorm" not in n)
            l2 = wd * tf.add_n([tf.nn.l2_loss(v) for v in tf.trainable_variables() if not_bn(v.name)])
            total = model_loss + l2
            variables = {v.op.name: v for v in tf.global_variables()}
            missing = [k for k in list(p) + list(st) if k not in variables]
            assert not missing, f"the oracle's variable names are not the graph's: {missing[:4]}"
            assign = [tf.assign(variables[k], np.asarray(v, np.float32).reshape(variables[k].shape.as_list())) for k, v in {**p, **st}.items()]
            with tf.Session(config=tf.ConfigProto(device_count={"GPU": 0})) as sess:
                if not is_training:
                    sess.run(tf.global_variables_initializer())
                    sess.run(assign)
                    lg, pr, rg = sess.run([logits, probs, endpoints["ranges"]], {inp: x})
                    res.update({"tf:eval_logits": lg, "tf:eval_probs": pr, "tf:eval_ranges": rg})
                    continue
                grads = tf.gradients(total, [variables[k] for k in p])
                opt = tf.train.MomentumOptimizer(learning_rate=lr, momentum=mu)
                gstep = tf.train.get_or_create_global_step()
                train_op = slim.learning.create_train_op(total, opt, global_step=gstep)      # runs the BN update ops (slim)
                sess.run(tf.global_variables_initializer())
                sess.run(assign)
                lg, ml, l2v, gv = sess.run([logits, model_loss, l2, grads], {inp: x, lab: labels})
                res.update({"tf:train_logits_keep1": lg, "tf:train_model_loss_keep1": ml, "tf:train_l2_loss": l2v})
                res.update({"tf:grad_keep1:" + k: g for k, g in zip(p, gv)})
                for step in (1, 2, 3):
                    sess.run(train_op, {inp: x, lab: labels})
                    if step in (1, 3):
                        vals = sess.run({k: variables[k] for k in list(p) + list(st)})
                        res.update({f"tf:param{step}_keep1:" + k: vals[k] for k in p})
                        res.update({f"tf:stat{step}_keep1:" + k: vals[k] for k in st})
        fx.update({k: np.asarray(v, np.float64) for k, v in res.items()})
        np.savez_compressed(os.path.join(out, fname), **fx)
        print("pinned", fname, len(res), "tf:* keys")


def pin_dscnn(tf, golden: str, out: str):
    """Adds `tf:*` keys to dscnn_4020.npz (eval, S / M / L) and dscnn_train_4020.npz (size S: train-mode forward, gradients, Adam steps);
    tests/test_oracle.py::test_dscnn_oracle_against_pinned_reference compares oracle/dscnn_ref.py with them."""
    slim = tf.contrib.slim
    import dataclasses
    from audio_nets import ds_cnn
    from oracle import dscnn_ref as D
    from oracle import numpy_ref as R
    defs = {"S": ds_cnn.S_NET_DEF, "M": ds_cnn.M_NET_DEF, "L": ds_cnn.L_NET_DEF}

    def build(size, x_shape, n_classes, is_training):
        tf.reset_default_graph()
        inp = tf.placeholder(tf.float32, [None] + list(x_shape))
        lab = tf.placeholder(tf.float32, [None, n_classes])
        with slim.arg_scope(ds_cnn.DSCNN_arg_scope(is_training=is_training)):
            logits, _ = ds_cnn.DSCNN(inp, n_classes, defs[size])
        return inp, lab, logits

    def assign_ops(values):
        variables = {v.op.name: v for v in tf.global_variables()}
        missing = [k for k in values if k not in variables]
        assert not missing, f"the oracle's variable names are not the graph's: {missing[:4]}"
        return variables, [tf.assign(variables[k], np.asarray(v, np.float32).reshape(variables[k].shape.as_list())) for k, v in values.items()]

    fx = dict(np.load(os.path.join(golden, "dscnn_4020.npz")))
    x = fx["mfcc"].astype(np.float32)[..., None]                              # [B, 49, 10, 1]
    res = {}
    for size in ("S", "M", "L"):
        p, st = D.init_params(D.net_def(size), seed=int(fx["init_seed"]))
        inp, lab, logits = build(size, x.shape[1:], 12, False)
        _, assign = assign_ops({**p, **st})
        with tf.Session(config=tf.ConfigProto(device_count={"GPU": 0})) as sess:
            sess.run(tf.global_variables_initializer())
            sess.run(assign)
            lg, pr = sess.run([logits, slim.softmax(logits)], {inp: x})
        res.update({f"tf:logits_{size}": lg, f"tf:probs_{size}": pr})
    fx.update({k: np.asarray(v, np.float64) for k, v in res.items()})
    np.savez_compressed(os.path.join(out, "dscnn_4020.npz"), **fx)
    print("pinned dscnn_4020.npz", sorted(res))

    tr = dict(np.load(os.path.join(golden, "dscnn_train_4020.npz")))
    p, st = D.init_params(D.net_def("S"), seed=int(tr["init_seed"]))
    wav = R.synth_waveforms(tr["labels"].shape[0], seed=int(tr["S:wav_seed"]))
    xt = R.mfcc(wav, dataclasses.replace(R.FRONTEND_4020, num_mfccs=10)).astype(np.float32)[..., None]     # (pin_frontend pins the MFCC itself)
    labels = tr["labels"].astype(np.float32)
    inp, lab, logits = build("S", xt.shape[1:], labels.shape[1], True)
    loss = tf.losses.softmax_cross_entropy(logits=logits, onehot_labels=lab, label_smoothing=0.0, weights=1.0)    # weight_decay 0.0 (factory/audio_nets.py:305)
    variables, assign = assign_ops({**p, **st})
    grads = tf.gradients(loss, [variables[k] for k in p])
    opt = tf.train.AdamOptimizer(learning_rate=5e-4)
    train_op = slim.learning.create_train_op(loss, opt, global_step=tf.train.get_or_create_global_step())     # runs the BN update ops (slim)
    res = {}
    with tf.Session(config=tf.ConfigProto(device_count={"GPU": 0})) as sess:
        sess.run(tf.global_variables_initializer())
        sess.run(assign)
        lg, gv = sess.run([logits, grads], {inp: xt, lab: labels})
        res["tf:S:train_logits"] = lg
        res.update({"tf:S:grad:" + k: g for k, g in zip(p, gv)})
        for step in (1, 2, 3):
            sess.run(train_op, {inp: xt, lab: labels})
            if step in (1, 3):
                vals = sess.run({k: variables[k] for k in list(p) + list(st)})
                res.update({f"tf:S:param{step}:" + k: vals[k] for k in p})
                res.update({f"tf:S:stat{step}:" + k: vals[k] for k in st})
    tr.update({k: np.asarray(v, np.float64) for k, v in res.items()})
    np.savez_compressed(os.path.join(out, "dscnn_train_4020.npz"), **tr)
    print("pinned dscnn_train_4020.npz", len(res), "tf:* keys")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True, help="tests/golden of this repository (fixtures are rewritten in place)")
    ap.add_argument("--only", default="all", choices=["frontend", "all"])
    args = ap.parse_args()
    tf = _require_reference()
    pin_frontend(tf, args.out, args.out)
    if args.only == "all":
        pin_nets(tf, args.out, args.out)
        pin_dscnn(tf, args.out, args.out)


if __name__ == "__main__":
    sys.exit(main())
