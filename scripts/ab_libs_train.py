#!/usr/bin/env python
"""Training-step A/B between this build and a side library (scripts/whatif_libs/$BASE, default lib_base.so), alternating in one process:
median step time by HIP events, batch 4096, features precomputed.  NETS=8,14 FRAMES=49,98 ROUNDS=4"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import tcresnet_amd as T
from bench import synth_batch
dev = torch.device("cuda")
libs = {"this": T._lib.get(), "base": T._lib.load_from(os.path.join(ROOT, "scripts", "whatif_libs", os.environ.get("BASE", "lib_base.so")), "hip")}
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
    return sorted(ts)[len(ts) // 2]
for fr in os.environ.get("FRAMES", "49").split(","):
    for nn in os.environ.get("NETS", "8,14").split(","):
        name, ch = ("TCResNet8", [16, 24, 32, 48]) if nn == "8" else ("TCResNet14", [24, 36, 36, 48, 48, 72, 72])
        nets, feats = {}, {}
        for k, lib in libs.items():
            fe = T.Frontend(window_size_samples=640 if fr == "49" else 480, window_stride_samples=320 if fr == "49" else 160, lib=lib, device=dev)
            feats[k] = fe(wav)
            nets[k] = T.TCResNet(name, ch, 40, fe.n_frames, 12, lib=lib, device=dev); nets[k].init_xavier(0)
        res = {k: [] for k in libs}
        for rnd in range(int(os.environ.get("ROUNDS", "4"))):
            for k in libs:
                net, feat = nets[k], feats[k]
                def train():
                    net.forward_train(feat, lab, keep_prob=0.5, seed=1); net.backward(); net.sgd_momentum_step(0.1, 0.9, 0.001)
                res[k].append(timeit(train))
        g = {}
        for k in libs:
            nets[k].init_xavier(0); nets[k].forward_train(feats[k], lab, keep_prob=0.5, seed=1); g[k] = nets[k].backward().clone()
        print(f"{name} {fr} frames: " + "  ".join(f"{k} {min(v):.0f} (median {sorted(v)[len(v) // 2]:.0f})" for k, v in res.items()),
              " gradients bitwise equal:", bool(torch.equal(g["this"], g["base"])), flush=True)
