#!/usr/bin/env python
"""Front-end kernel A/B between this build and a side library of another revision (scripts/whatif_libs/lib_fe_base.so), alternating in one
process, batch 4096: median launch time by HIP events after a warm-up, both front-end settings; outputs compared bitwise."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import tcresnet_amd as T
from bench import synth_batch
dev = torch.device("cuda")
wav = synth_batch(4096, dev, 1234)
libs = {"this": T._lib.get(), "base": T._lib.load_from(os.path.join(ROOT, "scripts", "whatif_libs", os.environ.get("BASE", "lib_fe_base.so")), "hip")}
for win, hop, coef in ((640, 320, 40), (480, 160, 40), (640, 320, 10)):
    fes = {k: T.Frontend(window_size_samples=win, window_stride_samples=hop, num_mfccs=coef, lib=l, device=dev) for k, l in libs.items()}
    outs = {k: fe(wav).clone() for k, fe in fes.items()}
    res = {k: [] for k in fes}
    for rnd in range(4):
        for k, fe in fes.items():
            buf = outs[k]
            for _ in range(60): fe(wav, out=buf)
            torch.cuda.synchronize()
            ts = []
            for _ in range(100):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); fe(wav, out=buf); e1.record(); e1.synchronize()
                ts.append(e0.elapsed_time(e1) * 1e3)
            res[k].append(sorted(ts)[50])
    print(f"{win}/{hop} x{coef}: " + "  ".join(f"{k} {min(v):.1f} us" for k, v in res.items()), " bitwise equal:", bool(torch.equal(outs["this"], outs["base"])), flush=True)
