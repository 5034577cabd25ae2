"""A/B of TCR_TUNE_WGRAD_TILES (output-channel tiles per launch of the 9-tap filter gradients) on the TC-ResNet training steps."""
import os, sys
sys.path.insert(0, "/root/repo")
import torch, tcresnet_amd as T
from bench import synth_batch
dev = torch.device("cuda"); lib = T._lib.get()
B = 4096
wav = synth_batch(B, dev, 1234)
lab = torch.zeros((B, 12), device=dev); lab[torch.arange(B), torch.arange(B) % 12] = 1
fe = T.Frontend(window_size_samples=640, window_stride_samples=320, device=dev)
feat = fe(wav)
def timeit(fn, n=40, warm=10):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort(); return ts[len(ts) // 2]
for name, ch in (("TCResNet14", [24, 36, 36, 48, 48, 72, 72]), ("TCResNet8", [16, 24, 32, 48])):
    net = T.TCResNet(name, ch, 40, fe.n_frames, 12, device=dev); net.init_xavier(0)
    def train():
        net.forward_train(feat, lab, keep_prob=0.5, seed=1); net.backward(); net.sgd_momentum_step(0.1, 0.9, 0.001)
    for rnd in range(2):
        for knob in (0, 3, 2):
            lib.tcr_tune(16, knob)
            print(f"{name} 9-tap filter gradient, <= {knob} channel tiles per launch: {timeit(train):9.1f} us", flush=True)
    lib.tcr_tune(16, 0)
