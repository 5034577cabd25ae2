#!/usr/bin/env python
"""A/B of the weight-gradient knobs on the TC-ResNet training step (features precomputed; fwd + bwd + momentum only)."""
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


def timeit(fn, n=20, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]

for name, ch in () if os.environ.get("AB_DS_ONLY") else (("TCResNet8", [16, 24, 32, 48]), ("TCResNet14", [24, 36, 36, 48, 48, 72, 72]), ("TCResNet8", [24, 36, 48, 72])):
    ref = None
    for cap in (0, 1, 2, 4):
        for unroll in (0,):
            lib.tcr_tune(6, cap)
            net = T.TCResNet(name, ch, 40, fe.n_frames, 12, device=dev)
            net.init_xavier(0)
            net.forward_train(feat, lab, keep_prob=0.5, seed=1); g = net.backward().clone()
            if cap == 0: ref = g
            same = bool(torch.equal(g, ref)) if cap == 0 else None
            def train():
                net.forward_train(feat, lab, keep_prob=0.5, seed=1); net.backward(); net.sgd_momentum_step(0.1, 0.9, 0.001)
            print(f"{name} ksplit={cap}: {timeit(train):9.1f} us  bitwise_vs_auto={same} maxdiff={float((g - ref).abs().max()):.3e}", flush=True)
            del net
lib.tcr_tune(6, 0)
fe3 = T.Frontend(window_size_samples=640, window_stride_samples=320, num_mfccs=10, device=dev)
feat3 = fe3(wav)
ds = T.DSCNN("L", fe3.n_frames, 10, 12, device=dev); ds.init_xavier(0)
st = [0]
def train_ds():
    st[0] += 1
    ds.forward_train(feat3, lab); ds.backward(); ds.adam_step(5e-4, st[0])
print(f"DSCNN-L train: {timeit(train_ds, n=6, warm=2):9.1f} us")
print(f"DSCNN-L eval forward: {timeit(lambda: ds.forward_infer(feat3), n=10, warm=3):9.1f} us")
