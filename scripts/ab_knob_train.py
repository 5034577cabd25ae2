#!/usr/bin/env python
"""A/B of one tcr_tune knob on the TC-ResNet training steps (batch 4096, features precomputed), values alternating in one process.
   KNOB=18 VALUES=0,404,208 NETS=8,14 FRAMES=49 python scripts/ab_knob_train.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import tcresnet_amd as T
from bench import synth_batch

dev = torch.device("cuda")
lib = T._lib.get()
KNOB = int(os.environ.get("KNOB", "18"))
VALUES = [int(v) for v in os.environ.get("VALUES", "0").split(",")]
NETS = os.environ.get("NETS", "8,14").split(",")
FRAMES = os.environ.get("FRAMES", "49").split(",")
ROUNDS = int(os.environ.get("ROUNDS", "2"))
for kv in filter(None, os.environ.get("TUNE", "").split(",")):
    lib.tcr_tune(int(kv.split("=")[0]), int(kv.split("=")[1]))
B = 4096
wav = synth_batch(B, dev, 1234)
lab = torch.zeros((B, 12), device=dev); lab[torch.arange(B), torch.arange(B) % 12] = 1


def timeit(fn, n=40, warm=10):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


for fr in FRAMES:
    fe = T.Frontend(window_size_samples=640 if fr == "49" else 480, window_stride_samples=320 if fr == "49" else 160, device=dev)
    feat = fe(wav)
    for nn in NETS:
        name, ch = ("TCResNet8", [16, 24, 32, 48]) if nn == "8" else ("TCResNet14", [24, 36, 36, 48, 48, 72, 72])
        net = T.TCResNet(name, ch, 40, fe.n_frames, 12, lib=lib, device=dev)
        net.init_xavier(0)
        def train():
            net.forward_train(feat, lab, keep_prob=0.5, seed=1); net.backward(); net.sgd_momentum_step(0.1, 0.9, 0.001)
        res = {}
        for rnd in range(ROUNDS):
            for v in VALUES:
                lib.tcr_tune(KNOB, v)
                res.setdefault(v, []).append(timeit(train))
        if os.environ.get("CHECK"):
            gs = {}
            for v in VALUES:
                lib.tcr_tune(KNOB, v)
                net.init_xavier(0)
                net.forward_train(feat, lab, keep_prob=0.5, seed=1)
                gs[v] = net.backward().clone()
            g0 = gs[VALUES[0]]
            print("   gradient max |diff| / max |g| vs first value: " + "  ".join(f"{v}: {float((gs[v] - g0).abs().max() / g0.abs().max()):.2e}" for v in VALUES[1:]))
        lib.tcr_tune(KNOB, 0)
        print(f"{name} {fr} frames: " + "  ".join(f"{v}: {min(t):.0f}" for v, t in res.items()), flush=True)
