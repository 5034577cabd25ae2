#!/usr/bin/env python
"""A/B of one tcr_tune knob on the eval forward of the network alone (batch 4096, features precomputed), values alternating in one
   process; the logits of every value are compared bitwise with the first's.
   KNOB=3 VALUES=0,7 NETS=8,14 FRAMES=49,98 python scripts/ab_knob_fwd.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import tcresnet_amd as T
from bench import synth_batch

dev = torch.device("cuda")
lib = T._lib.get()
KNOB = int(os.environ.get("KNOB", "3"))
VALUES = [int(v) for v in os.environ.get("VALUES", "0,7").split(",")]
NETS = os.environ.get("NETS", "8,14").split(",")
FRAMES = os.environ.get("FRAMES", "49,98").split(",")
ROUNDS = int(os.environ.get("ROUNDS", "3"))
B = int(os.environ.get("BATCH", "4096"))
wav = synth_batch(B, dev, 1234)


def timeit(fn, n=60, warm=15):
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
        res, outs = {}, {}
        for rnd in range(ROUNDS):
            for v in VALUES:
                lib.tcr_tune(KNOB, v)
                res.setdefault(v, []).append(timeit(lambda: net.forward_infer(feat)))
                outs[v] = net.forward_infer(feat)[0].clone()
        lib.tcr_tune(KNOB, 0)
        same = all(torch.equal(outs[v], outs[VALUES[0]]) for v in VALUES)
        print(f"{name} {fr} frames: " + "  ".join(f"{v}: {min(t):.1f}" for v, t in res.items()) + f"  bitwise {same}", flush=True)
