#!/usr/bin/env python
"""Front-end A/B: frames per workgroup (TCR_TUNE_FRONTEND = 10 + rounds; 0 = launcher policy), several batch sizes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import tcresnet_amd as T
from bench import synth_batch

dev = torch.device("cuda")
lib = T._lib.get()


def timeit(fn, n=30, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]

for B in (4096, 1024, 3000, 16384):
    wav = synth_batch(B, dev, 1234)
    for tag, win, hop, rr in (("4020", 640, 320, (0, 8, 7, 6, 5, 4, 0)), ("3010", 480, 160, (0, 4, 3, 2, 0))):
        fe = T.Frontend(window_size_samples=win, window_stride_samples=hop, device=dev)
        lib.tcr_tune(1, 0)
        ref = fe(wav).clone()
        out = torch.empty_like(ref)
        line = []
        for r in rr:
            lib.tcr_tune(1, 10 + r if r else 0)
            out.zero_()
            fe(wav, out=out)
            same = torch.equal(out, ref)
            line.append(f"r{r}: {timeit(lambda: fe(wav, out=out)):7.1f} us{'' if same else ' MISMATCH'}")
        lib.tcr_tune(1, 0)
        print(f"B={B} {tag}: " + "  ".join(line), flush=True)
