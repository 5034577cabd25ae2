#!/usr/bin/env python
"""DS-CNN-L / M training step with the register-staged pointwise filter gradient (TCR_TUNE_PW_WGRAD = 0, default) against the
DMA-staged kernel (1), alternating in one process; features precomputed, batch 4096."""
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
fe = T.Frontend(window_size_samples=640, window_stride_samples=320, num_mfccs=10, device=dev)
feat = fe(wav)


def timeit(fn, n=12, warm=4):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


for size in os.environ.get("AB_SIZES", "L,M").split(","):
    ds = T.DSCNN(size, fe.n_frames, 10, 12, device=dev); ds.init_xavier(0)
    st = [0]
    def step():
        st[0] += 1
        ds.forward_train(feat, lab); ds.backward(); ds.adam_step(5e-4, st[0])
    def bwd():
        ds.backward()
    step()
    for knob in (0, 1, 0, 1):
        lib.tcr_tune(25, knob)
        print(f"DSCNN-{size} pw_wgrad knob {knob}: backward {timeit(bwd):9.1f} us  step {timeit(step):9.1f} us", flush=True)
    lib.tcr_tune(25, 0)
