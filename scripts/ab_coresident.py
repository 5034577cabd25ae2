#!/usr/bin/env python
"""Co-resident front-end(k+1) || fused network(k): both kernels persistent, grids capped so that one workgroup of each fits a CU
(front-end 78.8 KB LDS + 246 VGPRs x 1 wave/SIMD, network 80.1 KB + 122 VGPRs x 2 waves/SIMD).  Wall-clock per step over N steps."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import tcresnet_amd as T
from tcresnet_amd.pipeline import InferencePipeline
from bench import synth_batch

dev = torch.device("cuda")
lib = T._lib.get()
B = int(os.environ.get("AB_BATCH", "4096"))
N = int(os.environ.get("AB_STEPS", "200"))
wav = synth_batch(B, dev, 1234)
FE_GRID, NET_GRID = 13, 14


def wall(fn, n=N, warm=50):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


for tag, win, hop in (("4020", 640, 320), ("3010", 480, 160)):
    fe = T.Frontend(window_size_samples=win, window_stride_samples=hop, device=dev)
    net = T.TCResNet("TCResNet8", [16, 24, 32, 48], 40, fe.n_frames, 12, device=dev)
    net.init_xavier(0)
    feat = fe(wav)
    out = (torch.empty((B, 12), device=dev), torch.empty((B, 12), device=dev))
    ref = net.forward_infer(feat)[0].clone()

    def seq():
        fe(wav, out=feat); net.forward_infer(feat, out=out)
    for fc, nc in ((0, 0), (256, 0), (0, 256), (256, 256)):
        lib.tcr_tune(FE_GRID, fc); lib.tcr_tune(NET_GRID, nc)
        print(f"{tag} sequential fe_cap {fc:4d} net_cap {nc:4d}: step {wall(seq):7.1f} us   fe alone {wall(lambda: fe(wav, out=feat)):7.1f}   net alone {wall(lambda: net.forward_infer(feat, out=out)):7.1f}", flush=True)
    for depth in (2, 3):
        for fc, nc in ((0, 0), (256, 256), (256, 512), (512, 256), (384, 256), (256, 384), (320, 256), (256, 320)):
            lib.tcr_tune(FE_GRID, fc); lib.tcr_tune(NET_GRID, nc)
            pipe = InferencePipeline(fe, net, B, depth=depth)
            def step():
                pipe.submit(wav)
            t = wall(step)
            pipe.sync(); torch.cuda.synchronize()
            ok = all(bool(torch.equal(o[0], ref)) for o in pipe.out)
            print(f"{tag} pipeline depth {depth} fe_cap {fc:4d} net_cap {nc:4d}: step {t:7.1f} us  bitwise {ok}", flush=True)
    lib.tcr_tune(FE_GRID, 0); lib.tcr_tune(NET_GRID, 0)
