#!/usr/bin/env python
"""Prefetched front-end next to a training step: does a smaller front-end grid (TCR_TUNE_FE_GRID: persistent workgroups; 0 = two per CU,
256 = one per CU, which leaves half of every CU's LDS to the step's kernels) cost the step less?  Batch 4096, front-end issued in front
of the forward on its own stream."""
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
        pf = FeaturePrefetcher(fe, B)
        pf.submit(wav)
        def step():
            f = pf.get()
            pf.submit(wav)
            net.forward_train(f, lab, keep_prob=0.5, seed=1); net.backward(); net.sgd_momentum_step(0.1, 0.9, 0.001)
        for rnd in range(2):
            for grid in (0, 384, 256, 128, 64):
                lib.tcr_tune(13, grid)
                print(f"{tag} {name} front-end grid cap {grid:4d}: step {timeit(step):8.1f} us", flush=True)
        lib.tcr_tune(13, 0)
        del net
