import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import tcresnet_amd as T
from bench import synth_batch
dev = torch.device("cuda"); B = 4096; lib = T._lib.get()
wav = synth_batch(B, dev, 1234)
fe = T.Frontend(window_size_samples=640, window_stride_samples=320, device=dev)
net = T.TCResNet("TCResNet8", [16, 24, 32, 48], 40, fe.n_frames, 12, device=dev); net.init_xavier(0)
feat = fe(wav)
for knob in (0, 6, 5):
    lib.tcr_tune(3, knob)
    for _ in range(12): net.forward_infer(feat)
    torch.cuda.synchronize()
