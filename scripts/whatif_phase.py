#!/usr/bin/env python
"""Times the TCResNet8 / 14 training FORWARD with each scripts/whatif_libs/lib_whatif_<mask>.so (train_fused.hip timing what-ifs; wrong results)."""
import glob, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import tcresnet_amd as T
from bench import synth_batch
dev = torch.device("cuda")
B = 4096
wav = synth_batch(B, dev, 1234)
lab = torch.zeros((B, 12), device=dev); lab[torch.arange(B), torch.arange(B) % 12] = 1
def timeit(fn, n=40, warm=10):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return sorted(ts)[len(ts) // 2]
libs = sorted(glob.glob(os.path.join(ROOT, "scripts", "whatif_libs", "lib_whatif_*.so")), key=lambda p: int(p.split("_")[-1][:-3]))
for path in libs:
    lib = T._lib.load_from(path, "hip")
    fe = T.Frontend(window_size_samples=640, window_stride_samples=320, lib=lib, device=dev)
    feat = fe(wav)
    out = []
    for name, ch in (("TCResNet8", [16, 24, 32, 48]), ("TCResNet14", [24, 36, 36, 48, 48, 72, 72])):
        net = T.TCResNet(name, ch, 40, fe.n_frames, 12, lib=lib, device=dev); net.init_xavier(0)
        out.append(f"{name} forward {timeit(lambda: net.forward_train(feat, lab, keep_prob=0.5, seed=1)):7.0f} us")
    print(f"mask {os.path.basename(path)[11:-3]:>3}: " + "   ".join(out), flush=True)
