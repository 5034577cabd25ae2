#!/usr/bin/env python
"""Experiment: capture one TC-ResNet training step (C-ABI launches on two streams) in a HIP graph via torch.cuda.CUDAGraph."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import tcresnet_amd as T
from bench import synth_batch

dev = torch.device("cuda")
B = 4096
wav = synth_batch(B, dev, 1234)
lab = torch.zeros((B, 12), device=dev); lab[torch.arange(B), torch.arange(B) % 12] = 1
fe = T.Frontend(window_size_samples=640, window_stride_samples=320, device=dev)
feat = fe(wav)


def timeit(fn, n=30, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]

for name, ch in (("TCResNet8", [16, 24, 32, 48]), ("TCResNet14", [24, 36, 36, 48, 48, 72, 72])):
    net = T.TCResNet(name, ch, 40, fe.n_frames, 12, device=dev)
    net.init_xavier(0)
    def train():
        net.forward_train(feat, lab, keep_prob=0.5, seed=1); net.backward(); net.sgd_momentum_step(0.1, 0.9, 0.001)
    print(f"{name} eager: {timeit(train):9.1f} us", flush=True)
    try:
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(3): train()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            train()
        torch.cuda.synchronize()
        p0 = net.params.clone() if hasattr(net, "params") else None
        print(f"{name} graph: {timeit(g.replay):9.1f} us", flush=True)
    except Exception as e:
        print(f"{name} graph capture failed: {type(e).__name__}: {str(e)[:300]}", flush=True)
