#!/usr/bin/env python
"""Training steps with each scripts/whatif_libs/lib_whatif_<tag>.so: TCResNet8 / TCResNet14-1.5 alone and with the next batch's front-end
prefetched (the bench's training legs), sets interleaved over ROUNDS."""
import glob, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import tcresnet_amd as T
from tcresnet_amd.pipeline import FeaturePrefetcher
from bench import synth_batch
dev = torch.device("cuda"); B = 4096
wav = synth_batch(B, dev, 1234)
lab = torch.zeros((B, 12), device=dev); lab[torch.arange(B), torch.arange(B) % 12] = 1
def timeit(fn, n=40, warm=10):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
libs = sorted(glob.glob(os.path.join(ROOT, "scripts", "whatif_libs", "lib_whatif_*.so")), key=lambda p: int(p.split("_")[-1][:-3]))
res = {}
for rnd in range(int(os.environ.get("ROUNDS", "2"))):
    for path in libs:
        lib = T._lib.load_from(path, "hip")
        fe = T.Frontend(window_size_samples=640, window_stride_samples=320, lib=lib, device=dev)
        feat = fe(wav)
        for name, ch in (("TCResNet8", [16, 24, 32, 48]), ("TCResNet14", [24, 36, 36, 48, 48, 72, 72])):
            net = T.TCResNet(name, ch, 40, fe.n_frames, 12, lib=lib, device=dev); net.init_xavier(0)
            def alone():
                net.forward_train(feat, lab, keep_prob=0.5, seed=1); net.backward(); net.sgd_momentum_step(0.1, 0.9, 0.001)
            pf = FeaturePrefetcher(fe, B); pf.submit(wav)
            def with_fe():
                f = pf.get(); pf.submit(wav, input_ready=True)
                net.forward_train(f, lab, keep_prob=0.5, seed=1); net.backward(); net.sgd_momentum_step(0.1, 0.9, 0.001)
            res.setdefault((os.path.basename(path)[11:-3], name), []).append((timeit(alone), timeit(with_fe)))
for (tag, name), r in res.items():
    print(f"lib {tag:>3} {name}: alone {min(x[0] for x in r):7.0f} us   with the next batch's front-end {min(x[1] for x in r):7.0f} us", flush=True)
