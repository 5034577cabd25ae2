#!/usr/bin/env python
"""Fused eval network, two 16-position tiles per job everywhere (default) against four in block 0's layers (TCR_TUNE_NET_FUSED = 5),
alternating in one process; TCResNet8-1.0 at 49 and 98 frames, batch 4096; outputs compared bitwise."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import tcresnet_amd as T
from bench import synth_batch
dev = torch.device("cuda"); B = 4096
lib = T._lib.get()
wav = synth_batch(B, dev, 1234)


def timeit(fn, n=100, warm=30):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


for win, hop in ((640, 320), (480, 160)):
    fe = T.Frontend(window_size_samples=win, window_stride_samples=hop, device=dev)
    net = T.TCResNet("TCResNet8", [16, 24, 32, 48], 40, fe.n_frames, 12, device=dev); net.init_xavier(0)
    feat = fe(wav)
    out = (torch.empty((B, 12), device=dev), torch.empty((B, 12), device=dev))
    res, outs = {0: [], 5: []}, {}
    for rnd in range(4):
        for knob in (0, 5):
            lib.tcr_tune(3, knob)
            res[knob].append(timeit(lambda: net.forward_infer(feat, out=out)))
            outs[knob] = out[0].clone()
    lib.tcr_tune(3, 0)
    print(f"{fe.n_frames} frames: two tiles per job: {min(res[0]):.1f} us   four in block 0: {min(res[5]):.1f} us   bitwise {bool(torch.equal(outs[0], outs[5]))}", flush=True)
