#!/usr/bin/env python
"""DS-CNN-L / M eval forward and training step with the wide pointwise conv kernel built for <= 128 registers (TCR_TUNE_PW_POS = 0,
default) against the unconstrained build (1: 92 VGPRs + 72 AGPRs, three waves per SIMD), alternating in one process."""
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
fe = T.Frontend(window_size_samples=640, window_stride_samples=320, num_mfccs=10, device=dev)
feat = fe(wav)


def timeit(fn, n=15, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


for size in os.environ.get("AB_SIZES", "L,M").split(","):
    st = [0]
    for rnd in range(2):
        for knob in tuple(int(x) for x in os.environ.get("AB_ARMS", "0,1").split(",")):
            lib.tcr_tune(28, knob)
            ds = T.DSCNN(size, fe.n_frames, 10, 12, device=dev); ds.init_xavier(0)
            def step():
                st[0] += 1
                ds.forward_train(feat, lab); ds.backward(); ds.adam_step(5e-4, st[0])
            step()
            print(f"DSCNN-{size} pw_pos {knob}: eval {timeit(lambda: ds.forward_infer(feat)):8.1f} us   train step {timeit(step):9.1f} us", flush=True)
            del ds
    lib.tcr_tune(28, 0)
