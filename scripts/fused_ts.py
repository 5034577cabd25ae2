"""Cycle counts between the barriers of the static fused eval kernel (debug build: fused.hip instrumented by hand)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, tcresnet_amd as T
from bench import synth_batch
dev = torch.device("cuda")
wav = synth_batch(4096, dev, 1234)
fe = T.Frontend(window_size_samples=640, window_stride_samples=320, device=dev)
net = T.TCResNet("TCResNet8", [16, 24, 32, 48], 40, fe.n_frames, 12, device=dev); net.init_xavier(0)
feat = fe(wav)
for _ in range(5): lg, pr = net.forward_infer(feat)
torch.cuda.synchronize()
lg, pr = net.forward_infer(feat)
torch.cuda.synchronize()
v = pr.flatten().cpu().numpy()
print("wg0  :", [int(x) for x in v[:12]])
print("wg300:", [int(x) for x in v[300 * 8 * 12 + 48: 300 * 8 * 12 + 60]])
