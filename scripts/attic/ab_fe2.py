#!/usr/bin/env python
"""Front-end, this build vs side libraries (scripts/build_ref_lib.py) in one process: time and bitwise equality."""
import glob, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import tcresnet_amd as T
from bench import synth_batch
dev = torch.device("cuda"); B = 4096
wav = synth_batch(B, dev, 1234)
def wall(fn, n=200, warm=50):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6
libs = [("this", T._lib.get())] + [(os.path.basename(p)[7:-3], T._lib.load_from(p, "hip", allow_missing=True)) for p in sorted(glob.glob(os.path.join(ROOT, "tc-resnet_amd", "lib", "side", "libtcr_*.so")))]
for tag, win, hop, nco in (("4020", 640, 320, 40), ("3010", 480, 160, 40), ("4020/10", 640, 320, 10)):
    fes = [(n, T.Frontend(window_size_samples=win, window_stride_samples=hop, num_mfccs=nco, lib=l, device=dev)) for n, l in libs]
    ref = fes[0][1](wav).clone()
    out = torch.empty_like(ref)
    for rep in range(2):
        for n, fe in fes:
            same = bool(torch.equal(fe(wav), ref))
            print(f"{tag:8s} {n:8s}: fe {wall(lambda: fe(wav, out=out)):7.1f} us  bitwise_vs_this {same}", flush=True)
