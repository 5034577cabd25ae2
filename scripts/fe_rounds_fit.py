"""Front-end kernel time vs rounds per workgroup (TCR_TUNE_FRONTEND = 10 + rounds): separates the per-round cost from the
per-workgroup fixed cost (table staging, twiddle loads, DCT, stores)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, tcresnet_amd as T
from bench import synth_batch
dev = torch.device("cuda")
lib = T._lib.get()
for B in (4096, 1024, 16384):
    wav = synth_batch(B, dev, 1234)
    fe = T.Frontend(window_size_samples=640, window_stride_samples=320, device=dev)
    out = fe(wav)
    for _ in range(60): fe(wav, out=out)
    for rounds in (0, 8, 7, 6, 5, 4):
        lib.tcr_tune(1, 10 + rounds if rounds else 0)
        for _ in range(10): fe(wav, out=out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(40): fe(wav, out=out)
        e1.record(); e1.synchronize()
        t = e0.elapsed_time(e1) / 40 * 1e3
        wgs = -(-B * 49 // ((rounds or 8) * 8))
        print(f"B={B} rounds={rounds}: {t:7.1f} us  workgroups {wgs} = {wgs / 512:.2f} dispatch waves", flush=True)
    lib.tcr_tune(1, 0)
