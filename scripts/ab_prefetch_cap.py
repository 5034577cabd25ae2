#!/usr/bin/env python
"""Training step with the next batch's front-end prefetched on its own stream, under caps on the front-end's persistent grid
(TCR_TUNE_FE_GRID: 0 = two workgroups per CU; 256 = one per CU, which leaves half of every CU's LDS / registers to the step's kernels)
and submit points.  CAPS=0,384,256,128 python scripts/ab_prefetch_cap.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import tcresnet_amd as T
from tcresnet_amd.pipeline import FeaturePrefetcher
from bench import synth_batch

dev = torch.device("cuda")
lib = T._lib.get()
B = 4096
CAPS = [int(v) for v in os.environ.get("CAPS", "0,384,256,128").split(",")]
wav = synth_batch(B, dev, 1234)
lab = torch.zeros((B, 12), device=dev); lab[torch.arange(B), torch.arange(B) % 12] = 1


def timeit(fn, n=60, warm=15):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


fe = T.Frontend(window_size_samples=640, window_stride_samples=320, device=dev)
for name, ch in (("TCResNet8", [16, 24, 32, 48]), ("TCResNet14", [24, 36, 36, 48, 48, 72, 72])):
    net = T.TCResNet(name, ch, 40, fe.n_frames, 12, device=dev)
    net.init_xavier(0)
    for point in ("none", "before_fwd", "after_fwd"):
        res = {}
        for rnd in range(2):
            for cap in CAPS:
                pf = FeaturePrefetcher(fe, B)
                pf.submit(wav)

                def step():
                    f = pf.feat[0] if point == "none" else pf.get()
                    lib.tcr_tune(13, cap)
                    if point == "before_fwd": pf.submit(wav)
                    lib.tcr_tune(13, 0)
                    net.forward_train(f, lab, keep_prob=0.5, seed=1)
                    lib.tcr_tune(13, cap)
                    if point == "after_fwd": pf.submit(wav)
                    lib.tcr_tune(13, 0)
                    net.backward()
                    net.sgd_momentum_step(0.1, 0.9, 0.001)
                res.setdefault(cap, []).append(timeit(step))
                if point == "none": break
        print(f"{name} front-end issued {point:10s}: " + "  ".join(f"cap {c}: {min(t):.0f}" for c, t in res.items()) + " us", flush=True)
