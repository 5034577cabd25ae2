#!/usr/bin/env python
"""DS-CNN A/B on one GPU: wide pointwise convs on the LDS-tiled kernel (default) vs the register-fed one (TCR_TUNE_CONV_B = 3)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import tcresnet_amd as T
from bench import synth_batch

dev = torch.device("cuda")
lib = T._lib.get()
B = int(os.environ.get("AB_BATCH", "4096"))
wav = synth_batch(B, dev, 1234)
lab = torch.zeros((B, 12), device=dev); lab[torch.arange(B), torch.arange(B) % 12] = 1
fe3 = T.Frontend(window_size_samples=640, window_stride_samples=320, num_mfccs=10, device=dev)
feat3 = fe3(wav)


def timeit(fn, n=8, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]

for size in os.environ.get("AB_SIZES", "L,M").split(","):
    ds = T.DSCNN(size, fe3.n_frames, 10, 12, device=dev); ds.init_xavier(0)
    st = [0]
    def train_ds():
        st[0] += 1
        ds.forward_train(feat3, lab); ds.backward(); ds.adam_step(5e-4, st[0])
    outs = {}
    for knob in (3, 0, 3, 0):
        lib.tcr_tune(2, knob)
        lg = ds.forward_infer(feat3)
        outs[knob] = (lg[0] if isinstance(lg, tuple) else lg).clone()
        print(f"DSCNN-{size} conv_b={knob}: eval {timeit(lambda: ds.forward_infer(feat3)):9.1f} us   train {timeit(train_ds, n=5, warm=2):9.1f} us", flush=True)
    lib.tcr_tune(2, 0)
    print(f"  max |logit diff| lds vs register-fed: {float((outs[0] - outs[3]).abs().max()):.3e}")
