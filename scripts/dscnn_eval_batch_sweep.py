#!/usr/bin/env python
"""DS-CNN-L eval forward (net only, one stream) at a few batch sizes around 4096: does the pointwise conv's fifth, 6 %-full round of
workgroups (4160 tiles of 64 positions on 1024 slots at batch 4096) show?  4032 utterances are 4095 tiles = four rounds."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import tcresnet_amd as T
from bench import synth_batch
dev = torch.device("cuda")
fe = T.Frontend(window_size_samples=640, window_stride_samples=320, num_mfccs=10, device=dev)
for B in [int(x) for x in (sys.argv[1:] or ["4096", "4032", "3968", "4096"])]:
    feat = fe(synth_batch(B, dev, 1234))
    ds = T.DSCNN("L", fe.n_frames, 10, 12, device=dev); ds.init_xavier(0)
    for _ in range(8): ds.forward_infer(feat)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30): ds.forward_infer(feat)
    e1.record(); e1.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 30
    print(f"batch {B}: {us:8.1f} us per batch = {us / B * 1e3:7.2f} ns per utterance", flush=True)
    del ds
