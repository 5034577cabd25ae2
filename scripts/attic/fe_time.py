"""Front-end kernel time at batch 4096 (HIP events), both flagship windows."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, tcresnet_amd as T
from bench import synth_batch
dev = torch.device("cuda")
wav = synth_batch(4096, dev, 1234)
LIB = T._lib.load_from(os.environ["TCR_LIB"], "hip") if os.environ.get("TCR_LIB") else None      # A/B against another build of the library
for tag, win, hop, kw in (("4020 mfcc40", 640, 320, {}), ("3010 mfcc40", 480, 160, {}), ("4020 mfcc10", 640, 320, {"num_mfccs": 10})):
    fe = T.Frontend(window_size_samples=win, window_stride_samples=hop, device=dev, lib=LIB, **kw)
    out = fe(wav)
    res = []
    for rep in range(3):
        for _ in range(5): fe(wav, out=out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30): fe(wav, out=out)
        e1.record(); e1.synchronize()
        res.append(e0.elapsed_time(e1) / 30 * 1e3)
    print(f"{tag}: " + " ".join(f"{r:7.1f}" for r in res) + " us", flush=True)
