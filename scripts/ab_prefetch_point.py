#!/usr/bin/env python
"""Where in the training step should the next batch's front-end be issued?  The train-mode forward runs as LDS-resident phases
(40-60 KB per workgroup) which cannot share a CU with two front-end workgroups (2 x 77 KB); the backward's kernels hold no LDS.
A: submit before the forward (round-1 order), B: submit after the forward (overlaps the backward), C: after the backward."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import tcresnet_amd as T
from tcresnet_amd.pipeline import FeaturePrefetcher
from bench import synth_batch

dev = torch.device("cuda")
B = int(os.environ.get("B", "4096"))
wav = synth_batch(B, dev, 1234)
lab = torch.zeros((B, 12), device=dev); lab[torch.arange(B), torch.arange(B) % 12] = 1


def timeit(fn, n=40, warm=10):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


for tag, win, hop in (("4020", 640, 320), ("3010", 480, 160)):
    fe = T.Frontend(window_size_samples=win, window_stride_samples=hop, device=dev)
    for name, ch in (("TCResNet8", [16, 24, 32, 48]), ("TCResNet14", [24, 36, 36, 48, 48, 72, 72])):
        net = T.TCResNet(name, ch, 40, fe.n_frames, 12, device=dev)
        net.init_xavier(0)
        for point in ("none", "before_fwd", "after_fwd", "after_bwd", "serial"):
            pf = FeaturePrefetcher(fe, B, overlap=point != "serial")
            pf.submit(wav)

            def step():
                if point == "none":
                    f = pf.feat[0]
                else:
                    f = pf.get()
                if point in ("before_fwd", "serial"): pf.submit(wav)
                net.forward_train(f, lab, keep_prob=0.5, seed=1)
                if point == "after_fwd": pf.submit(wav)
                net.backward()
                if point == "after_bwd": pf.submit(wav)
                net.sgd_momentum_step(0.1, 0.9, 0.001)
            t = timeit(step)
            print(f"{tag} {name}-{ch[0]} front-end issued {point:10s}: step {t:8.1f} us", flush=True)
        del net
