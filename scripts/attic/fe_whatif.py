"""Timing-only what-ifs of the packed front-end kernel (wrong results). bits: 1 mel items, 2 log + item sums, 4 real-FFT split,
256 log phase entirely, 8 second radix-16 pass, 16 first radix-16 pass, 64 inter-pass twiddles, 32 sample loads, 128 DCT MFMAs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, tcresnet_amd as T
from bench import synth_batch
dev = torch.device("cuda")
wav = synth_batch(4096, dev, 1234)
fe = T.Frontend(window_size_samples=640, window_stride_samples=320, device=dev)
out = fe(wav)
for _ in range(100): fe(wav, out=out)
for w in (0, 1, 2, 256, 258, 4, 8, 16, 64, 32, 128, 259, 263, 511, 0):
    os.environ["TCR_FE_WHATIF"] = str(w)
    for _ in range(8): fe(wav, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30): fe(wav, out=out)
    e1.record(); e1.synchronize()
    print(f"whatif {w:3d}: {e0.elapsed_time(e1) / 30 * 1e3:7.1f} us", flush=True)
