#!/usr/bin/env python
"""Times the what-if side libraries of the fused eval network (scripts/build_whatif.py).  Results of masks != 0 are WRONG by design."""
import glob, os, re, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import tcresnet_amd as T
from bench import synth_batch
dev = torch.device("cuda"); B = 4096
wav = synth_batch(B, dev, 1234)
fe = T.Frontend(window_size_samples=640, window_stride_samples=320, device=dev)
feat = fe(wav)
NAMES = {1: "no MFMAs", 2: "weights of tap 0 only", 4: "LDS operands of tap 0 only", 8: "no barriers", 16: "no head", 32: "no epilogue stores", 64: "no first conv"}
def wall(fn, n=200, warm=50):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6
libs = sorted(glob.glob(os.path.join(ROOT, "tc-resnet_amd", "lib", "whatif", "libtcr_w*.so")), key=lambda p: int(re.findall(r"_w(\d+)\.so", p)[0]))
for so in libs:
    m = int(re.findall(r"_w(\d+)\.so", so)[0])
    lib = T._lib.load_from(so, "hip")
    net = T.TCResNet("TCResNet8", [16, 24, 32, 48], 40, fe.n_frames, 12, lib=lib, device=dev); net.init_xavier(0)
    out = (torch.empty((B, 12), device=dev), torch.empty((B, 12), device=dev))
    t = wall(lambda: net.forward_infer(feat, out=out))
    print(f"mask {m:3d}: {t:7.1f} us   " + (" + ".join(v for k, v in NAMES.items() if m & k) or "product kernel"), flush=True)
