#!/usr/bin/env python
"""Forward step (front-end + fused network) eager vs HIP-graph replay (1 / 4 steps per graph) vs two-stream pipeline: wall-clock per step."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import tcresnet_amd as T
from tcresnet_amd.pipeline import InferencePipeline
from bench import synth_batch
dev = torch.device("cuda"); B = 4096
wav = synth_batch(B, dev, 1234)
fe = T.Frontend(window_size_samples=640, window_stride_samples=320, device=dev)
net = T.TCResNet("TCResNet8", [16, 24, 32, 48], 40, fe.n_frames, 12, device=dev); net.init_xavier(0)
feat = torch.empty((B, 40, fe.n_frames + 8), device=dev)
out = (torch.empty((B, 12), device=dev), torch.empty((B, 12), device=dev))
def step():
    fe(wav, out=feat); net.forward_infer(feat, out=out)
def wall(fn, n, warm=50, per=1):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / (n * per) * 1e6
for _ in range(100): step()
print(f"eager: {wall(step, 400):7.1f} us/step")
ref = out[0].clone()
for per in (1, 4, 16):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        step()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(per): step()
    out[0].zero_()
    t = wall(g.replay, 400 // per, warm=20, per=per)
    torch.cuda.synchronize()
    print(f"graph of {per:2d} steps: {t:7.1f} us/step  bitwise {bool(torch.equal(out[0], ref))}")
print(f"eager again: {wall(step, 400):7.1f} us/step")
for depth in (2, 3):
    pipe = InferencePipeline(fe, net, B, depth=depth)
    print(f"two-stream pipeline depth {depth}: {wall(lambda: pipe.submit(wav), 400):7.1f} us/step")
    pipe.sync(); torch.cuda.synchronize()
