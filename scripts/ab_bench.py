#!/usr/bin/env python
"""A/B micro-benchmarks on one GPU: per-piece event timings under the tcr_tune knobs (same process, interleaved)."""
import os, sys, json, time
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


def timeit(fn, n=20, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2], ts[0]

res = {}
for tag, win, hop in (("4020", 640, 320), ("3010", 480, 160)):
    fe = T.Frontend(window_size_samples=win, window_stride_samples=hop, device=dev)
    feat = fe(wav)
    for knob in (1, 4, 5):
        lib.tcr_tune(1, knob)
        res[f"frontend_{tag}_var{knob}_us"] = timeit(lambda: fe(wav, out=feat))
    lib.tcr_tune(1, 0)
    for name, ch in () if os.environ.get("AB_FE_ONLY") else (("TCResNet8", [16, 24, 32, 48]),) if os.environ.get("AB_QUICK") else (("TCResNet8", [16, 24, 32, 48]), ("TCResNet14w15", [24, 36, 36, 48, 48, 72, 72])):
        net = T.TCResNet("TCResNet8" if name == "TCResNet8" else "TCResNet14", ch, 40, fe.n_frames, 12, device=dev)
        net.init_xavier(0)
        for path in (0, 2):
            lib.tcr_tune(0, 0); lib.tcr_tune(2, 1 if path == 2 else 0)
            res[f"net_infer_{name}_{tag}_conv{path}_us"] = timeit(lambda: net.forward_infer(feat))
            def train():
                net.forward_train(feat, lab, keep_prob=0.5, seed=1); net.backward(); net.sgd_momentum_step(0.1, 0.9, 0.001)
            res[f"net_train_{name}_{tag}_conv{path}_us"] = timeit(train, n=10, warm=3)
        lib.tcr_tune(0, 0); lib.tcr_tune(2, 0)
for k, v in res.items():
    print("%-46s median %9.1f  min %9.1f" % (k, v[0], v[1]))
