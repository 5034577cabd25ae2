#!/usr/bin/env python
"""Training-step A/B between THIS build and another revision's library (scripts/build_ref_lib.py <ref> prev), alternating in one
process: TCResNet8-1.0 and TCResNet14-1.5, batch 4096, features precomputed (fwd + bwd + momentum)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import tcresnet_amd as T
from bench import synth_batch

dev = torch.device("cuda")
libs = {"head": T._lib.get(), "prev": T._lib.load_from(os.path.join(ROOT, "tc-resnet_amd", "lib", "side", "libtcr_prev.so"), "hip", allow_missing=True)}
B = 4096
wav = synth_batch(B, dev, 1234)
lab = torch.zeros((B, 12), device=dev); lab[torch.arange(B), torch.arange(B) % 12] = 1
fe = T.Frontend(window_size_samples=640, window_stride_samples=320, device=dev)
feat = fe(wav)


def timeit(fn, n=40, warm=10):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


for name, ch in (("TCResNet8", [16, 24, 32, 48]), ("TCResNet14", [24, 36, 36, 48, 48, 72, 72])) if os.environ.get("AB_TC", "1") == "1" else ():
    nets, grads = {}, {}
    for k, lib in libs.items():
        net = T.TCResNet(name, ch, 40, fe.n_frames, 12, lib=lib, device=dev)
        net.init_xavier(0)
        net.forward_train(feat, lab, keep_prob=0.5, seed=1)
        grads[k] = net.backward().clone()
        nets[k] = net
    print(f"{name}: gradients bitwise equal across revisions: {bool(torch.equal(grads['head'], grads['prev']))}")
    for rnd in range(3):
        for k, net in nets.items():
            def train():
                net.forward_train(feat, lab, keep_prob=0.5, seed=1); net.backward(); net.sgd_momentum_step(0.1, 0.9, 0.001)
            print(f"  {name} {k}: {timeit(train):9.1f} us", flush=True)

if os.environ.get("AB_DS", "1") == "1":
    fe3 = T.Frontend(window_size_samples=640, window_stride_samples=320, num_mfccs=10, device=dev)
    feat3 = fe3(wav)
    dsn = {}
    for k, lib in libs.items():
        ds = T.DSCNN("L", fe3.n_frames, 10, 12, lib=lib, device=dev); ds.init_xavier(0)
        dsn[k] = ds
    for rnd in range(3):
        for k, ds in dsn.items():
            st = [0]
            def train_ds():
                st[0] += 1
                ds.forward_train(feat3, lab); ds.backward(); ds.adam_step(5e-4, st[0])
            print(f"  DSCNN-L {k}: train step {timeit(train_ds, n=10, warm=3):9.1f} us   eval forward {timeit(lambda: ds.forward_infer(feat3), n=10, warm=3):9.1f} us", flush=True)
