#!/usr/bin/env python
"""A/B of the fused eval network: compile-time-specialised TCResNet8 kernel (default) vs the generic layer walk (TCR_TUNE_NET_FUSED = 3)."""
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


for tag, win, hop in (("4020", 640, 320), ("3010", 480, 160)):
    fe = T.Frontend(window_size_samples=win, window_stride_samples=hop, device=dev)
    feat = fe(wav)
    net = T.TCResNet("TCResNet8", [16, 24, 32, 48], 40, fe.n_frames, 12, device=dev); net.init_xavier(0)
    ref = None
    for name, knobs in (("generic g8 w8", {3: 3}), ("static g8 w8", {}), ("static w4", {5: 404}), ("static w16", {5: 416}), ("static g4", {4: 4}),
                        ("static g4 w4", {4: 4, 5: 404}), ("static g6", {4: 6}), ("static g16 w16", {4: 16, 5: 416})):
        for k, v in knobs.items(): lib.tcr_tune(k, v)
        try:
            out = net.forward_infer(feat)[0].clone()
            if ref is None: ref = out
            t = timeit(lambda: net.forward_infer(feat))
            print(f"{tag} {name:16s}: {t:8.1f} us  bitwise={bool(torch.equal(out, ref))}", flush=True)
        except Exception as e:
            print(f"{tag} {name}: {e}")
        for k in knobs: lib.tcr_tune(k, 0)
