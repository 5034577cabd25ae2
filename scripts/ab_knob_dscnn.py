#!/usr/bin/env python
"""A/B of one tcr_tune knob on DS-CNN-L (batch 4096): eval forward and the training step, values alternating in one process.
   KNOB=23 VALUES=1,0 python scripts/ab_knob_dscnn.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import tcresnet_amd as T
from bench import synth_batch
dev = torch.device("cuda"); lib = T._lib.get()
KNOB = int(os.environ.get("KNOB", "23")); VALUES = [int(v) for v in os.environ.get("VALUES", "1,0").split(",")]
B = 4096
wav = synth_batch(B, dev, 1234)
lab = torch.zeros((B, 12), device=dev); lab[torch.arange(B), torch.arange(B) % 12] = 1
def timeit(fn, n=12, warm=4):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1))
    return sorted(ts)[len(ts) // 2]
fe = T.Frontend(window_size_samples=640, window_stride_samples=320, num_mfccs=10, device=dev)
feat = fe(wav)
ds = T.DSCNN(os.environ.get("SIZE", "L"), fe.n_frames, 10, 12, device=dev); ds.init_xavier(0)
def train():
    ds.forward_train(feat, lab); ds.backward(); ds.adam_step(1e-3, 1)
res = {}
for rnd in range(int(os.environ.get("ROUNDS", "3"))):
    for v in VALUES:
        lib.tcr_tune(KNOB, v)
        res.setdefault(v, []).append((timeit(lambda: ds.forward_infer(feat)), timeit(train, 8, 3)))
lib.tcr_tune(KNOB, 0)
for v, r in res.items():
    print(f"knob {KNOB} = {v}: eval {min(x[0] for x in r):.3f} ms   train {min(x[1] for x in r):.3f} ms", flush=True)
