"""Sequential vs two-stream (InferencePipeline) eval forward at batch 4096: front-end(k+1) || network(k)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, tcresnet_amd as T
from tcresnet_amd.pipeline import InferencePipeline
from bench import synth_batch
dev = torch.device("cuda"); lib = T._lib.get()
B = 4096
wav = synth_batch(B, dev, 1234)
fe = T.Frontend(window_size_samples=640, window_stride_samples=320, device=dev)
net = T.TCResNet("TCResNet8", [16, 24, 32, 48], 40, fe.n_frames, 12, device=dev); net.init_xavier(0)
feat = torch.empty((B, 40, fe.n_frames + 8), device=dev)
out = (torch.empty((B, 12), device=dev), torch.empty((B, 12), device=dev))
def seq():
    fe(wav, out=feat); net.forward_infer(feat, out=out)
def run(fn, n=60, warm=10, sync=None):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    if sync: sync()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
print(f"sequential: {run(seq):.4f} ms/step")
for g, w in ((0, 0), (4, 404), (4, 408), (8, 408), (2, 404)):
    lib.tcr_tune(4, g); lib.tcr_tune(5, w)
    pipe = InferencePipeline(fe, net, B)
    print(f"pipeline group={g} knob={w}: {run(lambda: pipe.submit(wav), sync=pipe.sync):.4f} ms/step   (sequential with the same knobs: {run(seq):.4f})")
lib.tcr_tune(4, 0); lib.tcr_tune(5, 0)
