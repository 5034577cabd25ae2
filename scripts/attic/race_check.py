#!/usr/bin/env python
"""Race detector for the multi-stream backward: the same training forward + backward repeated N times at batch 4096 must give bitwise
the same gradients every time (TCResNet8-1.0, TCResNet14-1.5, DS-CNN-L)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, tcresnet_amd as T
from bench import synth_batch
dev = torch.device("cuda")
B, N = 4096, int(os.environ.get("N", "40"))
wav = synth_batch(B, dev, 1234)
lab = torch.zeros((B, 12), device=dev); lab[torch.arange(B), torch.arange(B) % 12] = 1
fe = T.Frontend(window_size_samples=640, window_stride_samples=320, device=dev)
feat = fe(wav)
fe3 = T.Frontend(window_size_samples=640, window_stride_samples=320, num_mfccs=10, device=dev)
feat3 = fe3(wav)
nets = [("TCResNet8", T.TCResNet("TCResNet8", [16, 24, 32, 48], 40, fe.n_frames, 12, device=dev), feat),
        ("TCResNet14-1.5", T.TCResNet("TCResNet14", [24, 36, 36, 48, 48, 72, 72], 40, fe.n_frames, 12, device=dev), feat),
        ("DSCNN-L", T.DSCNN("L", fe3.n_frames, 10, 12, device=dev), feat3)]
for name, net, f in nets:
    net.init_xavier(0)
    stats0 = net.stats.clone()
    ref = None
    bad = 0
    for i in range(N):
        net.stats.copy_(stats0)
        if name.startswith("TC"):
            lg, _, loss = net.forward_train(f, lab, keep_prob=0.5, seed=7)
        else:
            lg, _, loss = net.forward_train(f, lab)
        g = net.backward().clone()
        if ref is None: ref = (lg.clone(), g)
        elif not (torch.equal(lg, ref[0]) and torch.equal(g, ref[1])): bad += 1
    print(f"{name}: {N} repetitions, {bad} differing from the first", flush=True)
    assert bad == 0
