#!/usr/bin/env python
"""Does a net's training step depend on how many nets (each with its own side streams) were created before it in the process?
HIP maps streams onto a few hardware queues; streams that share a queue serialise.  Batch 4096, TCResNet8, features precomputed."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import tcresnet_amd as T
from bench import synth_batch

dev = torch.device("cuda")
lib = T._lib.get()
B = 4096
wav = synth_batch(B, dev, 1234)
lab = torch.zeros((B, 12), device=dev); lab[torch.arange(B), torch.arange(B) % 12] = 1
fe = T.Frontend(window_size_samples=640, window_stride_samples=320, device=dev)
feat = fe(wav)


def timeit(fn, n=60, warm=15):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


pre = [torch.cuda.Stream(dev) for _ in range(int(os.environ.get("PRE_STREAMS", "0")))]     # streams other code created earlier
for st in pre:
    with torch.cuda.stream(st): torch.zeros(1, device=dev)
torch.cuda.synchronize()
nets = []
for i in range(int(os.environ.get("NETS", "6"))):
    net = T.TCResNet("TCResNet8", [16, 24, 32, 48], 40, fe.n_frames, 12, lib=lib, device=dev)
    net.init_xavier(0)
    nets.append(net)
    def train(net=net):
        net.forward_train(feat, lab, keep_prob=0.5, seed=1); net.backward(); net.sgd_momentum_step(0.1, 0.9, 0.001)
    print(f"net #{i} (created {i + 1}th): {timeit(train):8.1f} us", flush=True)
for i, net in enumerate(nets):
    def train(net=net):
        net.forward_train(feat, lab, keep_prob=0.5, seed=1); net.backward(); net.sgd_momentum_step(0.1, 0.9, 0.001)
    print(f"again net #{i}: {timeit(train):8.1f} us", flush=True)
