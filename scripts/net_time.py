"""Eval network time at batch 4096 (HIP events)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, tcresnet_amd as T
from bench import synth_batch
dev = torch.device("cuda")
LIB = T._lib.load_from(os.environ["TCR_LIB"], "hip") if os.environ.get("TCR_LIB") else None
wav = synth_batch(4096, dev, 1234)
fe = T.Frontend(window_size_samples=640, window_stride_samples=320, device=dev, lib=LIB)
net = T.TCResNet("TCResNet8", [16, 24, 32, 48], 40, fe.n_frames, 12, device=dev, lib=LIB); net.init_xavier(0)
feat = fe(wav)
res = []
for rep in range(3):
    for _ in range(10): net.forward_infer(feat)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(40): net.forward_infer(feat)
    e1.record(); e1.synchronize()
    res.append(e0.elapsed_time(e1) / 40 * 1e3)
print(os.environ.get("TCR_LIB", "default"), " ".join(f"{r:7.1f}" for r in res), "us", flush=True)
