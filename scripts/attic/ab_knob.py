#!/usr/bin/env python
"""A/B of one tune knob on the TCResNet8 and DS-CNN-L training steps: KNOB=<index> VALUES=a,b,..."""
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


def timeit(fn, n=40, warm=10):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


fe = T.Frontend(window_size_samples=640, window_stride_samples=320, device=dev)
feat = fe(wav)
net = T.TCResNet("TCResNet8", [16, 24, 32, 48], 40, fe.n_frames, 12, device=dev); net.init_xavier(0)
fe3 = T.Frontend(window_size_samples=640, window_stride_samples=320, num_mfccs=10, device=dev)
feat3 = fe3(wav)
ds = T.DSCNN("L", fe3.n_frames, 10, 12, device=dev); ds.init_xavier(0)
k = [0]


def ds_step():
    k[0] += 1
    ds.forward_train(feat3, lab); ds.backward(); ds.adam_step(1e-3, k[0])


def tc_step():
    net.forward_train(feat, lab, keep_prob=0.5, seed=1); net.backward(); net.sgd_momentum_step(0.1, 0.9, 0.001)


knob = int(os.environ["KNOB"])
for rep in range(2):
    for v in [int(x) for x in os.environ["VALUES"].split(",")]:
        lib.tcr_tune(knob, v)
        print(f"knob {knob} = {v}: TCResNet8 step {timeit(tc_step):8.1f} us   DS-CNN-L step {timeit(ds_step, n=6, warm=2):9.1f} us", flush=True)
lib.tcr_tune(knob, 0)
