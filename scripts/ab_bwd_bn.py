#!/usr/bin/env python
"""A/B of the BN backward: finalize folded into the apply pass (TCR_TUNE_BWD_BN_FUSED = 0 / workgroup targets) vs the finalize + apply
pair (1).  Features precomputed; times forward + backward + momentum."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import tcresnet_amd as T
from bench import synth_batch

dev = torch.device("cuda")
lib = T._lib.get()
B = int(os.environ.get("B", "4096"))
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


for tag, win, hop in (("4020", 640, 320), ("3010", 480, 160)):
    fe = T.Frontend(window_size_samples=win, window_stride_samples=hop, device=dev)
    feat = fe(wav)
    for name, ch in (("TCResNet8", [16, 24, 32, 48]), ("TCResNet14", [24, 36, 36, 48, 48, 72, 72])):
        net = T.TCResNet(name, ch, 40, fe.n_frames, 12, device=dev)
        net.init_xavier(0)

        def step():
            net.forward_train(feat, lab, keep_prob=0.5, seed=1); net.backward(); net.sgd_momentum_step(0.1, 0.9, 0.001)
        res = []
        for knob in [int(k) for k in os.environ.get("KNOBS", "1,0,1024,2048").split(",")]:
            lib.tcr_tune(11, knob)
            res.append(f"{knob}: {timeit(step):7.1f}")
        print(f"{tag} {name}-{ch[0]} step us by knob  " + "  ".join(res), flush=True)
        del net
lib.tcr_tune(11, 0)

# ReLU masks of the BN backward recomputed from the raw conv outputs (TCR_TUNE_BWD_MASK = 0) vs read back from the activations (1)
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


for mask in (1, 0, 1, 0):
    lib.tcr_tune(12, mask)
    print(f"mask knob {mask}: TCResNet8 step {timeit(tc_step):8.1f} us   DS-CNN-L step {timeit(ds_step, n=6, warm=2):9.1f} us", flush=True)
lib.tcr_tune(12, 0)
