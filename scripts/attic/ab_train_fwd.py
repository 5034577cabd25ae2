#!/usr/bin/env python
"""A/B of the train-mode forward: group-resident phases (TCR_TUNE_TRAIN_FWD = 0) vs the per-layer kernels (1).
Features precomputed; times forward only, and forward + backward + momentum."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import tcresnet_amd as T
from bench import synth_batch

dev = torch.device("cuda")
lib = T._lib.get()
B = int(os.environ.get("B", "4096"))
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


for tag, win, hop in (("4020", 640, 320), ("3010", 480, 160)):
    fe = T.Frontend(window_size_samples=win, window_stride_samples=hop, device=dev)
    feat = fe(wav)
    for name, ch in (("TCResNet8", [16, 24, 32, 48]), ("TCResNet14", [24, 36, 36, 48, 48, 72, 72])):
        outs = {}
        for knob, bknob in ((1, 0), (0, 0), (0, 1)):
            lib.tcr_tune(8, knob)
            lib.tcr_tune(9, bknob)
            net = T.TCResNet(name, ch, 40, fe.n_frames, 12, device=dev)
            net.init_xavier(0)
            lg, _, loss = net.forward_train(feat, lab, keep_prob=0.5, seed=1)
            g = net.backward().clone()
            outs[(knob, bknob)] = (lg.clone(), g, net.stats.clone())
            tf = timeit(lambda: net.forward_train(feat, lab, keep_prob=0.5, seed=1))

            def train():
                net.forward_train(feat, lab, keep_prob=0.5, seed=1); net.backward(); net.sgd_momentum_step(0.1, 0.9, 0.001)
            tt = timeit(train)
            tb = timeit(lambda: (net.forward_train(feat, lab, keep_prob=0.5, seed=1), net.backward()))
            print(f"{tag} {name}-{ch[0]} fwd={'phases' if knob == 0 else 'per-layer'} bwd={'phases' if bknob == 1 else 'per-layer'}: forward {tf:8.1f} us   fwd+bwd {tb:8.1f} us   step {tt:8.1f} us", flush=True)
            del net
        d = [float((a - b).abs().max()) for a, b in zip(outs[(0, 1)], outs[(1, 0)])]
        print(f"   max |phases - per-layer|: logits {d[0]:.2e} grads {d[1]:.2e} (max |g| {float(outs[(1, 0)][1].abs().max()):.2e}) stats {d[2]:.2e}", flush=True)
lib.tcr_tune(8, 0)
lib.tcr_tune(9, 0)
