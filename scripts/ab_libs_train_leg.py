#!/usr/bin/env python
"""The bench's training LEG (every step computes its own MFCC, prefetched on a second stream: bench.py train_leg) A/B between this build and a side
library (scripts/whatif_libs/$BASE, default lib_base.so), alternating in one process: wall time per step over 100 steps.  NETS=8,14 FRAMES=49 ROUNDS=4"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import tcresnet_amd as T
from tcresnet_amd.pipeline import FeaturePrefetcher
from bench import synth_batch
dev = torch.device("cuda")
libs = {"this": T._lib.get(), "base": T._lib.load_from(os.path.join(ROOT, "scripts", "whatif_libs", os.environ.get("BASE", "lib_base.so")), "hip")}
B = 4096
wav = synth_batch(B, dev, 1234)
lab = torch.zeros((B, 12), device=dev); lab[torch.arange(B), torch.arange(B) % 12] = 1
for fr in os.environ.get("FRAMES", "49").split(","):
    for nn in os.environ.get("NETS", "8,14").split(","):
        name, ch = ("TCResNet8", [16, 24, 32, 48]) if nn == "8" else ("TCResNet14", [24, 36, 36, 48, 48, 72, 72])
        steps = {}
        for k, lib in libs.items():
            fe = T.Frontend(window_size_samples=640 if fr == "49" else 480, window_stride_samples=320 if fr == "49" else 160, lib=lib, device=dev)
            net = T.TCResNet(name, ch, 40, fe.n_frames, 12, lib=lib, device=dev); net.init_xavier(0)
            pf = FeaturePrefetcher(fe, B)
            pf.submit(wav)
            cnt = [0]
            def step(net=net, pf=pf, cnt=cnt):
                cnt[0] += 1
                f = pf.get()
                pf.submit(wav, input_ready=True)
                net.forward_train(f, lab, keep_prob=0.5, seed=cnt[0]); net.backward(); net.sgd_momentum_step(0.1, 0.9, 0.001)
            steps[k] = step
        res = {k: [] for k in libs}
        for rnd in range(int(os.environ.get("ROUNDS", "4"))):
            for k in libs:
                for _ in range(30): steps[k]()
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for _ in range(100): steps[k]()
                torch.cuda.synchronize()
                res[k].append((time.perf_counter() - t0) / 100 * 1e6)
        print(f"{name} {fr} frames, leg with the prefetched front-end: " + "  ".join(f"{k} {min(v):.0f} (median {sorted(v)[len(v) // 2]:.0f})" for k, v in res.items()), flush=True)
