#!/usr/bin/env python
"""DS-CNN training step A/B on one GPU: normalised activations never materialised (TCR_TUNE_DS_TRAIN = 0, default) vs the
materialising path (1); 2: lazy, but every BN backward by a bn_bwd_apply pass.  Times forward, backward and the whole step (features precomputed)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import tcresnet_amd as T
from bench import synth_batch

dev = torch.device("cuda")
lib = T._lib.get()
B = int(os.environ.get("AB_BATCH", "4096"))
wav = synth_batch(B, dev, 1234)
lab = torch.zeros((B, 12), device=dev); lab[torch.arange(B), torch.arange(B) % 12] = 1
fe = T.Frontend(window_size_samples=640, window_stride_samples=320, num_mfccs=10, device=dev)
feat = fe(wav)


def timeit(fn, n=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


for size in os.environ.get("AB_SIZES", "L,M").split(","):
    ds = T.DSCNN(size, fe.n_frames, 10, 12, device=dev); ds.init_xavier(0)
    st = [0]
    def step():
        st[0] += 1
        ds.forward_train(feat, lab); ds.backward(); ds.adam_step(5e-4, st[0])
    def fwd():
        ds.forward_train(feat, lab)
    def bwd():
        ds.backward()
    step()
    for knob in [int(k) for k in os.environ.get("AB_KNOBS", "1,2,4,0,1,2,4,0").split(",")]:
        lib.tcr_tune(15, knob)
        t_f = timeit(fwd)
        t_b = timeit(bwd)
        t_s = timeit(step)
        print(f"DSCNN-{size} ds_train={knob}: forward {t_f:9.1f} us  backward {t_b:9.1f} us  step {t_s:9.1f} us", flush=True)
    lib.tcr_tune(15, 0)
