#!/usr/bin/env python
"""Per-kernel geometry sweep of the lazy backward (TCR_TUNE_BWD_LAZY_CFG addressed to one data-gradient kernel at a time): step time of
the TC-ResNet training step, batch 4096, for every (G, KS) of one kernel with the others on the cost model.  NET=8|14 FRAMES=49|98"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import tcresnet_amd as T
from bench import synth_batch
dev = torch.device("cuda"); lib = T._lib.get()
B = 4096
wav = synth_batch(B, dev, 1234)
lab = torch.zeros((B, 12), device=dev); lab[torch.arange(B), torch.arange(B) % 12] = 1
fr = os.environ.get("FRAMES", "49")
fe = T.Frontend(window_size_samples=640 if fr == "49" else 480, window_stride_samples=320 if fr == "49" else 160, device=dev)
feat = fe(wav)
nn = os.environ.get("NET", "8")
name, ch = ("TCResNet8", [16, 24, 32, 48]) if nn == "8" else ("TCResNet14", [24, 36, 36, 48, 48, 72, 72])
net = T.TCResNet(name, ch, 40, fe.n_frames, 12, lib=lib, device=dev); net.init_xavier(0)
def train():
    net.forward_train(feat, lab, keep_prob=0.5, seed=1); net.backward(); net.sgd_momentum_step(0.1, 0.9, 0.001)
def timeit(n=25, warm=6):
    for _ in range(warm): train()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); train(); e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return sorted(ts)[len(ts) // 2]
LAZY = int(os.environ.get("LAZY", "3"))        # 3: lazy forced (the default policy keeps nets of > 48 channels on the per-layer chain)
lib.tcr_tune(9, 2); ref = timeit(); lib.tcr_tune(9, LAZY)
base = timeit()
print(f"{name} {fr} frames: per-layer {ref:.0f} us, lazy (cost model) {base:.0f} us", flush=True)
# data-gradient kernels of the net: (out channels, layers): conv_b of block i -> (ch[i+1], 1); conv_a -> (ch[i], 2 with a shortcut conv else 1)
kernels = []
for i in range(len(ch) - 1):
    kernels.append((ch[i + 1], 1)); kernels.append((ch[i], 2 if ch[i] != ch[i + 1] else 1))
for oc, nl in sorted(set(kernels)):
    row = []
    for ks in (1, 2, 4, 8):
        for g in (1, 2, 3, 4, 5, 6, 8, 10, 12, 16):
            lib.tcr_tune(18, (oc * 10 + nl) * 10000 + ks * 100 + g)
            t = timeit(12, 3)
            if abs(t - ref) > 0.004 * ref: row.append((t, g, ks))     # (== per-layer: the forced geometry does not fit, the net fell back)
    lib.tcr_tune(18, 0)
    row.sort()
    print(f"  kernel out {oc} layers {nl}: best " + "  ".join(f"G{g}/KS{ks}: {t:.0f}" for t, g, ks in row[:6]), flush=True)
