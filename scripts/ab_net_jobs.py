#!/usr/bin/env python
"""Fused eval network, arms of TCR_TUNE_NET_FUSED (AB_ARMS, default 0,8,9: units + weight lookahead (default), jobs of two tiles (rounds 3-5), units only; 5: four tiles in block 0),
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


NETS = {"8": ("TCResNet8", [16, 24, 32, 48]), "14": ("TCResNet14", [24, 36, 36, 48, 48, 72, 72])}
for win, hop, which in ((640, 320, "8"), (480, 160, "8"), (640, 320, "14"), (480, 160, "14")):
    fe = T.Frontend(window_size_samples=win, window_stride_samples=hop, device=dev)
    net = T.TCResNet(NETS[which][0], NETS[which][1], 40, fe.n_frames, 12, device=dev); net.init_xavier(0)
    feat = fe(wav)
    out = (torch.empty((B, 12), device=dev), torch.empty((B, 12), device=dev))
    arms = tuple(int(x) for x in os.environ.get('AB_ARMS', '0,8,9').split(','))
    res, outs = {k: [] for k in arms}, {}
    for rnd in range(4):
        for knob in arms:
            lib.tcr_tune(3, knob)
            res[knob].append(timeit(lambda: net.forward_infer(feat, out=out)))
            outs[knob] = out[0].clone()
    lib.tcr_tune(3, 0)
    print(f"{NETS[which][0]} {fe.n_frames} frames: " + "   ".join(f"knob {k}: {min(res[k]):.1f} us" for k in arms) + f"   bitwise {all(bool(torch.equal(outs[arms[0]], outs[k])) for k in arms)}", flush=True)
