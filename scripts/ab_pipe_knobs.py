#!/usr/bin/env python
"""Alternating two-stream pipeline (bench.py's headline) under sets of tcr_tune knobs, sets interleaved in one process.
   SETS="|8=256|8=384|8=256,4=4" python scripts/ab_pipe_knobs.py      (each set: knob=value,...; empty = defaults)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import tcresnet_amd as T
from tcresnet_amd.pipeline import InferencePipeline
from bench import synth_batch
dev = torch.device("cuda")
lib = T._lib.get()
B = int(os.environ.get("BATCH", "4096"))
SETS = os.environ.get("SETS", "|8=256").split("|")
ROUNDS = int(os.environ.get("ROUNDS", "3"))
fe = T.Frontend(window_size_samples=640, window_stride_samples=320, device=dev)
net = T.TCResNet("TCResNet8", [16, 24, 32, 48], 40, fe.n_frames, 12, device=dev); net.init_xavier(0)
wav = synth_batch(B, dev, 1234)


def apply(spec, on=True):
    for kv in filter(None, spec.split(",")):
        lib.tcr_tune(int(kv.split("=")[0]), int(kv.split("=")[1]) if on else 0)


def run(pipe, n=200, warm=60):
    for _ in range(warm): pipe.submit(wav)
    pipe.sync(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): pipe.submit(wav)
    pipe.sync(); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


WAYS = [int(v) for v in os.environ.get("WAYS", "2").split(",")]
FE_ROUNDS = os.environ.get("FE_ROUNDS", "auto")
pipes = {w: InferencePipeline(fe, net, B, mode="alternate", ways=w, fe_rounds=(FE_ROUNDS if FE_ROUNDS == "auto" else int(FE_ROUNDS))) for w in WAYS}
res = {}
for rnd in range(ROUNDS):
    for spec in SETS:
        for w in WAYS:
            apply(spec)
            res.setdefault(f"{spec} ways={w}", []).append(run(pipes[w]))
            apply(spec, False)
for spec, ts in res.items():
    print(f"{spec or 'defaults':>24}: " + "  ".join(f"{t:.1f}" for t in ts) + " us per batch", flush=True)
