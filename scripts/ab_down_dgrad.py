#!/usr/bin/env python
"""A/B of TCR_TUNE_DOWN_DGRAD (0: a block's shortcut data gradient early on the side stream, conv_a's adds onto it; 1: behind conv_a's on
the main stream) on the TC-ResNet training steps, alternating in one process.  Batch 4096, features precomputed."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import tcresnet_amd as T
from bench import synth_batch

dev = torch.device("cuda")
lib = T._lib.get()
KNOB = int(os.environ.get("KNOB", "17"))                                  # 17: TCR_TUNE_DOWN_DGRAD (2 early for every net / 1 behind conv_a); 7: TCR_TUNE_WGRAD_STREAM (0 / 2: shortcut units on the second internal stream)
KNOBS = [int(v) for v in os.environ.get("VALUES", "2,1").split(",")]
B = 4096
wav = synth_batch(B, dev, 1234)
lab = torch.zeros((B, 12), device=dev); lab[torch.arange(B), torch.arange(B) % 12] = 1


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


for stride, tag in ((320, "49 frames"), (160, "98 frames")):
    fe = T.Frontend(window_size_samples=640 if stride == 320 else 480, window_stride_samples=stride, device=dev)
    feat = fe(wav)
    for name, ch in (("TCResNet8", [16, 24, 32, 48]), ("TCResNet14", [24, 36, 36, 48, 48, 72, 72])):
        net = T.TCResNet(name, ch, 40, fe.n_frames, 12, lib=lib, device=dev)
        net.init_xavier(0)
        def train():
            net.forward_train(feat, lab, keep_prob=0.5, seed=1); net.backward(); net.sgd_momentum_step(0.1, 0.9, 0.001)
        for rnd in range(3):
            for knob in KNOBS:
                lib.tcr_tune(KNOB, knob)
                print(f"  {name} {tag} knob {knob}: {timeit(train):9.1f} us", flush=True)
        lib.tcr_tune(KNOB, 0)
