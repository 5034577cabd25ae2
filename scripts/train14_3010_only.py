"""A few TCResNet14-1.5 training steps at 98 frames (30/10 ms: the reference's own script for this model) on precomputed features (profiling target)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, tcresnet_amd as T
from bench import synth_batch
dev = torch.device("cuda")
for kv in filter(None, os.environ.get("TUNE", "").split(",")):
    T._lib.get().tcr_tune(int(kv.split("=")[0]), int(kv.split("=")[1]))
B = 4096
wav = synth_batch(B, dev, 1234)
lab = torch.zeros((B, 12), device=dev); lab[torch.arange(B), torch.arange(B) % 12] = 1
fe = T.Frontend(window_size_samples=480, window_stride_samples=160, device=dev)
feat = fe(wav)
net = T.TCResNet("TCResNet14", [24, 36, 36, 48, 48, 72, 72], 40, fe.n_frames, 12, device=dev); net.init_xavier(0)
for i in range(int(os.environ.get("STEPS", "6"))):
    net.forward_train(feat, lab, keep_prob=0.5, seed=i); net.backward(); net.sgd_momentum_step(0.1, 0.9, 0.001)
torch.cuda.synchronize()
