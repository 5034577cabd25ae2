#!/usr/bin/env python
"""Times DS-CNN-L (batch 4096) eval forward and training step with each scripts/whatif_libs/lib_whatif_<mask>.so (timing what-ifs; wrong results)."""
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
def timeit(fn, n=12, warm=4):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1))
    return sorted(ts)[len(ts) // 2]
libs = sorted(glob.glob(os.path.join(ROOT, "scripts", "whatif_libs", "lib_whatif_*.so")), key=lambda p: int(p.split("_")[-1][:-3]))
for path in libs:
    lib = T._lib.load_from(path, "hip")
    fe = T.Frontend(window_size_samples=640, window_stride_samples=320, num_mfccs=10, lib=lib, device=dev)
    feat = fe(wav)
    ds = T.DSCNN("L", fe.n_frames, 10, 12, lib=lib, device=dev); ds.init_xavier(0)
    te = timeit(lambda: ds.forward_infer(feat))
    def train():
        ds.forward_train(feat, lab); ds.backward(); ds.adam_step(1e-3, 1)
    try:
        tt = timeit(train, 6, 2)
    except Exception as e:
        tt = float("nan")
    print(f"mask {os.path.basename(path)[11:-3]:>3}: eval {te:7.3f} ms   train {tt:7.3f} ms", flush=True)
