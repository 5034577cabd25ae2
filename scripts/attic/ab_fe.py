#!/usr/bin/env python
"""Front-end A/B: workgroup-chunk kernel of round 2 (knob 1 = 5) vs wave-autonomous chunks (0); bitwise check, several batch sizes."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import tcresnet_amd as T
from bench import synth_batch
dev = torch.device("cuda"); lib = T._lib.get()
B = int(os.environ.get("AB_BATCH", "4096"))
wav = synth_batch(B + 3, dev, 1234)
def wall(fn, n=200, warm=50):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6
KN = [int(k) for k in os.environ.get("AB_KNOBS", "5,0").split(",")]
for tag, win, hop, nco, meth in (("4020", 640, 320, 40, "mfcc"), ("3010", 480, 160, 40, "mfcc"), ("4020/10", 640, 320, 10, "mfcc"), ("4020 logmel", 640, 320, 40, "log_mel")):
    kw = dict(window_size_samples=win, window_stride_samples=hop, num_mfccs=nco, device=dev)
    try:
        fe = T.Frontend(method=meth, **kw) if meth != "mfcc" else T.Frontend(**kw)
    except TypeError:
        continue
    w = wav[:B].contiguous()
    lib.tcr_tune(1, 5); ref = fe(w).clone()
    out = torch.empty_like(ref)
    for knob in KN + KN:
        lib.tcr_tune(1, knob)
        o = fe(w)
        print(f"{tag:12s} knob {knob}: fe {wall(lambda: fe(w, out=out)):7.1f} us  bitwise_vs_round2 {bool(torch.equal(o, ref))}  maxdiff {float((o-ref).abs().max()):.2e}", flush=True)
    for bsz in (1, 3, 64, 1000, B + 3):
        ww = wav[:bsz].contiguous()
        lib.tcr_tune(1, 5); r = fe(ww).clone()
        lib.tcr_tune(1, 0); o = fe(ww)
        print(f"{tag:12s} batch {bsz}: bitwise {bool(torch.equal(o, r))}")
lib.tcr_tune(1, 0)
