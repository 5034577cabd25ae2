#!/usr/bin/env python
"""Fused eval network, this build vs side libraries (scripts/build_ref_lib.py) in one process: TCResNet8 (static + generic walk) and TCResNet14-1.5."""
import glob, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import tcresnet_amd as T
from bench import synth_batch
dev = torch.device("cuda"); B = 4096
wav = synth_batch(B, dev, 1234)
def wall(fn, n=100, warm=30):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6
libs = [("this", T._lib.get())] + [(os.path.basename(p)[7:-3], T._lib.load_from(p, "hip", allow_missing=True)) for p in sorted(glob.glob(os.path.join(ROOT, "tc-resnet_amd", "lib", "side", "libtcr_*.so")))]
for tag, win, hop in (("4020", 640, 320), ("3010", 480, 160)):
    fe = T.Frontend(window_size_samples=win, window_stride_samples=hop, device=dev)
    feat = fe(wav)
    for name, scope, ch, knob in (("TCResNet8 static", "TCResNet8", [16, 24, 32, 48], 0), ("TCResNet8 generic walk", "TCResNet8", [16, 24, 32, 48], 3),
                                  ("TCResNet14-1.5", "TCResNet14", [24, 36, 36, 48, 48, 72, 72], 0), ("TCResNet8-0.5 (12/18/24/36)", "TCResNet8", [8, 12, 16, 24], 0),
                                  ("TCResNet8-1.5", "TCResNet8", [24, 36, 48, 72], 0)):
        outs = {}
        for ln, lib in libs:
            net = T.TCResNet(scope, ch, 40, fe.n_frames, 12, lib=lib, device=dev); net.init_xavier(0)
            g = torch.Generator(device="cpu").manual_seed(3)
            for n, ti in net.tensors.items():
                if ti.kind in (1, 4): net._view(n).copy_((0.5 + torch.rand(tuple(net._view(n).shape), generator=g)).to(dev))
                elif ti.kind in (2, 3): net._view(n).copy_((torch.rand(tuple(net._view(n).shape), generator=g) - 0.5).to(dev))
            lib.tcr_tune(3, knob)
            o = net.forward_infer(feat, want_ranges=True)
            outs[ln] = [t.clone() for t in o]
            t = wall(lambda: net.forward_infer(feat))
            lib.tcr_tune(3, 0)
            same = all(bool(torch.equal(a, b)) for a, b in zip(outs[ln][:1], outs["this"][:1]))
            print(f"{tag} {name:30s} {ln:6s}: {t:8.1f} us  logits bitwise_vs_this {same}  probs maxdiff {float((outs[ln][1]-outs['this'][1]).abs().max()):.1e}  ranges maxdiff {float((outs[ln][2]-outs['this'][2]).abs().max()):.1e}", flush=True)
