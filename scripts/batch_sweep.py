"""Eval-forward latency / throughput of the default path over batch sizes (waveform -> softmax, TCResNet8-1.0, 49x40)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, tcresnet_amd as T
from bench import synth_batch
dev = torch.device("cuda")
fe = T.Frontend(window_size_samples=640, window_stride_samples=320, device=dev)
net = T.TCResNet("TCResNet8", [16, 24, 32, 48], 40, fe.n_frames, 12, device=dev); net.init_xavier(0)
ref_wav = synth_batch(64, dev, 1234)
ref = net.forward_infer(fe(ref_wav))[0].clone()
for B in (1, 3, 16, 64, 256, 1024, 4096, 16384, 65536):
    wav = synth_batch(B, dev, 1234)
    feat = torch.empty((B, 40, fe.n_frames + 8), device=dev)
    out = (torch.empty((B, 12), device=dev), torch.empty((B, 12), device=dev))
    def step():
        fe(wav, out=feat); net.forward_infer(feat, out=out)
    for _ in range(5): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 50 if B <= 4096 else 10
    for _ in range(n): step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    k = min(B, 64)
    same = torch.equal(out[0][:k], net.forward_infer(fe(wav[:k].contiguous()))[0])     # batch-size invariance of the first rows
    print(f"B={B:6d}: {dt * 1e6:9.1f} us/step  {B / dt / 1e6:7.3f} M utt/s  first rows identical to a small-batch run: {same}")
