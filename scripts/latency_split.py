#!/usr/bin/env python
"""Where the batch-1 latency goes: the Python wrapper (forward_waveform) vs a prepared C-ABI call (waveform_call), device vs stream
synchronisation, launch only (no synchronisation)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import tcresnet_amd as T
from bench import synth_batch
dev = torch.device("cuda")
wav = synth_batch(64, dev, 1234)
fe = T.Frontend(window_size_samples=640, window_stride_samples=320, device=dev)
net = T.TCResNet("TCResNet8", [16, 24, 32, 48], 40, fe.n_frames, 12, device=dev); net.init_xavier(0)
for b in (1, 8, 64):
    w1 = wav[:b].contiguous()
    o1 = (torch.empty((b, 12), device=dev), torch.empty((b, 12), device=dev))
    ref = net.forward_waveform(fe, w1)[0].clone()
    call = net.waveform_call(fe, w1, o1)
    st = torch.cuda.current_stream()
    def run(fn, sync, n=400):
        for _ in range(50): fn(); sync()
        t0 = time.perf_counter()
        for _ in range(n): fn(); sync()
        return (time.perf_counter() - t0) / n * 1e6
    a = run(lambda: net.forward_waveform(fe, w1, out=o1), torch.cuda.synchronize)
    c = run(call, torch.cuda.synchronize)
    d = run(call, st.synchronize)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(400): call()
    e = (time.perf_counter() - t0) / 400 * 1e6
    torch.cuda.synchronize()
    call(); torch.cuda.synchronize()
    print(f"batch {b}: forward_waveform + device sync {a:.1f} us | prepared call + device sync {c:.1f} | prepared call + stream sync {d:.1f} | "
          f"prepared call, no sync (host launch cost) {e:.1f} | bitwise {bool(torch.equal(o1[0], ref))}", flush=True)
