#!/usr/bin/env python
"""DS-CNN-L / M training step with one tuning knob at 0 (default) and at 1, alternating in one process.
usage: ab_dscnn_knob.py KNOB[=VALUE] [KNOB[=VALUE] ...]   (e.g. 29 = TCR_TUNE_BN_APPLY, 30 = TCR_TUNE_DW_DGRAD, 15=4; VALUE defaults to 1);
the last arm sets all listed knobs."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import tcresnet_amd as T
from bench import synth_batch

spec = [(int(x.split("=")[0]), int(x.split("=")[1]) if "=" in x else 1) for x in sys.argv[1:]] or [(29, 1), (30, 1)]
knobs = [k for k, _ in spec]
val = dict(spec)
dev = torch.device("cuda")
lib = T._lib.get()
B = 4096
wav = synth_batch(B, dev, 1234)
lab = torch.zeros((B, 12), device=dev); lab[torch.arange(B), torch.arange(B) % 12] = 1
fe = T.Frontend(window_size_samples=640, window_stride_samples=320, num_mfccs=10, device=dev)
feat = fe(wav)


def timeit(fn, n=15, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


for size in os.environ.get("AB_SIZES", "L,M").split(","):
    st = [0]
    arms = [()] + [(k,) for k in knobs] + ([tuple(knobs)] if len(knobs) > 1 else [])
    for rnd in range(2):
        for arm in arms:
            for k in knobs: lib.tcr_tune(k, val[k] if k in arm else 0)
            ds = T.DSCNN(size, fe.n_frames, 10, 12, device=dev); ds.init_xavier(0)
            def step():
                st[0] += 1
                ds.forward_train(feat, lab); ds.backward(); ds.adam_step(5e-4, st[0])
            step()
            print(f"DSCNN-{size} knobs at 1: {str(list(arm)):12s} train step {timeit(step):9.1f} us", flush=True)
            del ds
    for k in knobs: lib.tcr_tune(k, 0)
