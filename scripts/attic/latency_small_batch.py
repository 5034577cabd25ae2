#!/usr/bin/env python
"""Small-batch latency of the eval path: three host calls (front-end, network with cached fold) vs the single tcr_forward_waveform call."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import tcresnet_amd as T
from bench import synth_batch
dev = torch.device("cuda")
wav = synth_batch(4096, dev, 1234)
fe = T.Frontend(window_size_samples=640, window_stride_samples=320, device=dev)
net = T.TCResNet("TCResNet8", [16, 24, 32, 48], 40, fe.n_frames, 12, device=dev); net.init_xavier(0)
def lat(fn, n=300, warm=50):
    """per-call latency with a host sync after every call (what a serving loop sees) and back-to-back throughput time"""
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn(); torch.cuda.synchronize()
    a = (time.perf_counter() - t0) / n * 1e6
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return a, (time.perf_counter() - t0) / n * 1e6
for b in (1, 8, 64, 256, 1024, 4096):
    w = wav[:b].contiguous()
    out = (torch.empty((b, 12), device=dev), torch.empty((b, 12), device=dev))
    feat = torch.empty((b, 40, fe.n_frames + 8), device=dev)
    r = net.forward_infer(fe(w))[0].clone()
    o = net.forward_waveform(fe, w)[0]
    a3, t3 = lat(lambda: net.forward_infer(fe(w, out=feat), out=out))
    a1, t1 = lat(lambda: net.forward_waveform(fe, w, out=out))
    print(f"batch {b:5d}: two calls {a3:7.1f} us synced / {t3:7.1f} us back-to-back;  single call {a1:7.1f} / {t1:7.1f} us;  bitwise {bool(torch.equal(o, r))}", flush=True)
