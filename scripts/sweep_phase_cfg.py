#!/usr/bin/env python
"""Sweep of the training-phase geometry (TCR_TUNE_PHASE_CFG = waves * 100 + utterances per group) for forward and backward phases."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import tcresnet_amd as T
from bench import synth_batch
dev = torch.device("cuda")
lib = T._lib.get()
B = 4096
wav = synth_batch(B, dev, 1234)
lab = torch.zeros((B, 12), device=dev); lab[torch.arange(B), torch.arange(B) % 12] = 1


def timeit(fn, n=30, warm=8):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


for tag, win, hop, name, ch in (("4020", 640, 320, "TCResNet8", [16, 24, 32, 48]), ("4020", 640, 320, "TCResNet14", [24, 36, 36, 48, 48, 72, 72]),
                                ("3010", 480, 160, "TCResNet8", [16, 24, 32, 48])):
    fe = T.Frontend(window_size_samples=win, window_stride_samples=hop, device=dev)
    feat = fe(wav)
    for bwd in (0, 1):
        for cfg in (0, 808, 806, 804, 408, 406, 404, 403, 402):
            lib.tcr_tune(9, bwd); lib.tcr_tune(10, cfg)
            net = T.TCResNet(name, ch, 40, fe.n_frames, 12, device=dev); net.init_xavier(0)
            tf = timeit(lambda: net.forward_train(feat, lab, keep_prob=0.5, seed=1))
            tb = timeit(lambda: (net.forward_train(feat, lab, keep_prob=0.5, seed=1), net.backward()))
            print(f"{tag} {name}-{ch[0]} bwd={'phases' if bwd == 1 else 'per-layer'} cfg={cfg:4d}: forward {tf:8.1f} us  fwd+bwd {tb:8.1f} us", flush=True)
            del net
lib.tcr_tune(9, 0); lib.tcr_tune(10, 0)
