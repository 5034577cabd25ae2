#!/usr/bin/env python
"""Sequential step vs the two-stream inference pipelines (handoff 2 / 3 deep, alternate), batch 4096, TCResNet8, both front-ends."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import tcresnet_amd as T
from tcresnet_amd.pipeline import InferencePipeline
from bench import synth_batch

dev = torch.device("cuda")
B = int(os.environ.get("AB_BATCH", "4096"))
wav = synth_batch(B, dev, 1234)


def wall(fn, n=150, warm=50):
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
    for rnd in range(2):
        print(f"{tag} sequential           : {wall(seq):7.1f} us", flush=True)
        for mode, depth in (("handoff", 2), ("handoff", 3), ("alternate", 2)):
            pipe = InferencePipeline(fe, net, B, depth=depth, mode=mode)
            t = wall(lambda: pipe.submit(wav))
            pipe.sync(); torch.cuda.synchronize()
            ok = all(bool(torch.equal(o[0], ref)) for o in pipe.out)
            print(f"{tag} {mode:9s} depth {depth}   : {t:7.1f} us  bitwise {ok}", flush=True)
