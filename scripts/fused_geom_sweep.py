#!/usr/bin/env python
"""Fused eval network, geometry re-sweep on the round-6 kernel (units + weight lookahead): utterances per group x waves per workgroup,
TCResNet8-1.0 and TCResNet14-1.5 at 49 and 98 frames, batch 4096 (TCR_TUNE_FUSED_GROUP / TCR_TUNE_FUSED_WAVES; 0/0 = the policy)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, tcresnet_amd as T
from bench import synth_batch
dev = torch.device("cuda"); lib = T._lib.get()
B = 4096
wav = synth_batch(B, dev, 1234)


def timeit(fn, n=60, warm=20):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


for tag, win, hop in (("49", 640, 320), ("98", 480, 160)):
    fe = T.Frontend(window_size_samples=win, window_stride_samples=hop, device=dev)
    feat = fe(wav)
    for name, scope, ch in (("TCResNet8-1.0", "TCResNet8", [16, 24, 32, 48]), ("TCResNet14-1.5", "TCResNet14", [24, 36, 36, 48, 48, 72, 72])):
        net = T.TCResNet(scope, ch, 40, fe.n_frames, 12, device=dev); net.init_xavier(0)
        ref = net.forward_infer(feat)[0].clone()
        res = []
        for g in (0, 3, 4, 5, 6, 7, 8):
            for w in (0, 404, 408, 416):
                if (g == 0) != (w == 0): continue
                lib.tcr_tune(4, g); lib.tcr_tune(5, w)
                try:
                    out = net.forward_infer(feat)[0]; torch.cuda.synchronize()
                    ok = bool(torch.equal(out, ref))
                    res.append((f"g{g}/w{w % 100}", min(timeit(lambda: net.forward_infer(feat)) for _ in range(2)), ok))
                except Exception as e:
                    res.append((f"g{g}/w{w % 100}", float("nan"), False))
        lib.tcr_tune(4, 0); lib.tcr_tune(5, 0)
        print(f"{name} {tag} frames: " + "  ".join(f"{k}:{t:.1f}{'' if ok else '!'}" for k, t, ok in res), flush=True)
