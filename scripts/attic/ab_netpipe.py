#!/usr/bin/env python
"""Fused eval network A/B: rolled-tap kernel of round 2 (knob 3 = 4) vs the software-pipelined taps (0: ring 2, 5: ring 3); bitwise check."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import tcresnet_amd as T
from bench import synth_batch
dev = torch.device("cuda")
lib = T._lib.get()
B = int(os.environ.get("AB_BATCH", "4096"))
wav = synth_batch(B, dev, 1234)
def wall(fn, n=200, warm=50):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6
for tag, win, hop in (("4020", 640, 320), ("3010", 480, 160)):
    fe = T.Frontend(window_size_samples=win, window_stride_samples=hop, device=dev)
    net = T.TCResNet("TCResNet8", [16, 24, 32, 48], 40, fe.n_frames, 12, device=dev)
    net.init_xavier(0)
    # non-trivial BN statistics
    g = torch.Generator(device="cpu").manual_seed(3)
    for n, ti in net.tensors.items():
        if ti.kind in (1, 4): net._view(n).copy_((0.5 + torch.rand(tuple(net._view(n).shape), generator=g)).to(dev))
        elif ti.kind in (2, 3): net._view(n).copy_((torch.rand(tuple(net._view(n).shape), generator=g) - 0.5).to(dev))
    feat = fe(wav)
    out = (torch.empty((B, 12), device=dev), torch.empty((B, 12), device=dev))
    lib.tcr_tune(3, 4)
    ref = [t.clone() for t in net.forward_infer(feat)]
    for knob in (4, 0, 4, 0):
        lib.tcr_tune(3, knob)
        o = net.forward_infer(feat)
        same = bool(torch.equal(o[0], ref[0])) and bool(torch.equal(o[1], ref[1]))
        print(f"{tag} knob {knob}: net {wall(lambda: net.forward_infer(feat, out=out)):7.1f} us  bitwise_vs_round2 {same}  maxdiff {float((o[0]-ref[0]).abs().max()):.2e}", flush=True)
    for bsz in (1, 7, 100, 4099 if B >= 4099 else B - 3):
        bsz = min(bsz, B)
        lib.tcr_tune(3, 4); r = net.forward_infer(feat[:bsz].contiguous())[0].clone()
        lib.tcr_tune(3, 0); o = net.forward_infer(feat[:bsz].contiguous())[0]
        print(f"{tag} batch {bsz}: bitwise {bool(torch.equal(o, r))}")
    lib.tcr_tune(3, 0)
