"""One eval-network launch set per batch size for PMC passes (scripts/gpu_pmc2.sh)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import tcresnet_amd as T
from bench import synth_batch
dev = torch.device("cuda"); B = 4096
wav = synth_batch(B, dev, 1234)
fe = T.Frontend(window_size_samples=640, window_stride_samples=320, device=dev)
net = T.TCResNet("TCResNet8", [16, 24, 32, 48], 40, fe.n_frames, 12, device=dev); net.init_xavier(0)
feat = fe(wav)
for b in (8, 2048, 4096):
    f = feat[:b].contiguous()
    for _ in range(12): net.forward_infer(f)
    torch.cuda.synchronize()
