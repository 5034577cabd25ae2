#!/usr/bin/env python
"""Next batch's front-end under the training step on streams of different PRIORITY (the prefetch is off the critical path: a filler),
and in chunks of 1 .. 8 rounds (TCR_TUNE_FRONTEND = 10 + rounds: short-lived workgroups give the CUs back sooner)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import tcresnet_amd as T
from tcresnet_amd.pipeline import FeaturePrefetcher
from bench import synth_batch

dev = torch.device("cuda")
lib = T._lib.get()
hip = ctypes.CDLL("libamdhip64.so")
lo, hi = ctypes.c_int(), ctypes.c_int()
hip.hipDeviceGetStreamPriorityRange(ctypes.byref(lo), ctypes.byref(hi))
print("stream priority range: least", lo.value, "greatest", hi.value, flush=True)


def make_stream(prio):
    st = ctypes.c_void_p()
    rc = hip.hipStreamCreateWithPriority(ctypes.byref(st), 1, prio)        # 1 = hipStreamNonBlocking
    assert rc == 0, rc
    return torch.cuda.ExternalStream(st.value, device=dev)


B = 4096
wav = synth_batch(B, dev, 1234)
lab = torch.zeros((B, 12), device=dev); lab[torch.arange(B), torch.arange(B) % 12] = 1


def timeit(fn, n=60, warm=15):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


fe = T.Frontend(window_size_samples=640, window_stride_samples=320, device=dev)
for name, ch in (("TCResNet8", [16, 24, 32, 48]), ("TCResNet14", [24, 36, 36, 48, 48, 72, 72])):
    net = T.TCResNet(name, ch, 40, fe.n_frames, 12, device=dev)
    net.init_xavier(0)

    def run(st, rounds):
        pf = FeaturePrefetcher(fe, B)
        if st is not None: pf.stream = st
        pf.submit(wav)

        def step():
            f = pf.get()
            lib.tcr_tune(1, 10 + rounds if rounds else 0)
            pf.submit(wav)
            lib.tcr_tune(1, 0)
            net.forward_train(f, lab, keep_prob=0.5, seed=1)
            net.backward()
            net.sgd_momentum_step(0.1, 0.9, 0.001)
        return timeit(step)

    print(f"{name} library stream (the library probes its streams on first use): r0 {run(None, 0):.0f}  r8 {run(None, 8):.0f}  r2 {run(None, 2):.0f}", flush=True)
    for p in sorted({lo.value, 0, hi.value}):
        for trial in range(3):                  # (a new stream may land on a hardware queue one of the library's streams uses: several tries)
            st = make_stream(p)
            print(f"{name} stream of priority {p:2d}, try {trial}: r0 {run(st, 0):.0f}  r8 {run(st, 8):.0f}  r2 {run(st, 2):.0f}", flush=True)
    print(f"{name} library stream again: r0 {run(None, 0):.0f}", flush=True)
