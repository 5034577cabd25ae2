"""A few TCResNet8 train-mode forwards on precomputed features (profiling target)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, tcresnet_amd as T
from bench import synth_batch
dev = torch.device("cuda")
B = 4096
wav = synth_batch(B, dev, 1234)
lab = torch.zeros((B, 12), device=dev); lab[torch.arange(B), torch.arange(B) % 12] = 1
fe = T.Frontend(window_size_samples=640, window_stride_samples=320, device=dev)
feat = fe(wav)
T._lib.get().tcr_tune(8, int(os.environ.get("TRAIN_FWD", "0")))
net = T.TCResNet("TCResNet8", [16, 24, 32, 48], 40, fe.n_frames, 12, device=dev); net.init_xavier(0)
for i in range(int(os.environ.get("STEPS", "6"))):
    net.forward_train(feat, lab, keep_prob=0.5, seed=i)
    if os.environ.get("WITH_BWD"):
        net.backward(); net.sgd_momentum_step(0.1, 0.9, 0.001)
torch.cuda.synchronize()
