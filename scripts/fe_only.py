"""A few launches of the front-end kernel at batch 4096 (profiling target)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, tcresnet_amd as T
from bench import synth_batch
dev = torch.device("cuda")
wav = synth_batch(4096, dev, 1234)
if os.environ.get("KNOB23"): T._lib.get().tcr_tune(23, int(os.environ["KNOB23"]))
fe = T.Frontend(window_size_samples=int(os.environ.get("WIN", "640")), window_stride_samples=int(os.environ.get("HOP", "320")), device=dev)
out = fe(wav)
for _ in range(int(os.environ.get("STEPS", "10"))): fe(wav, out=out)
torch.cuda.synchronize()
