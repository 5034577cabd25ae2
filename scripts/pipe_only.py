#!/usr/bin/env python
"""Profiling target: the headline schedule -- whole batches alternating over three streams (InferencePipeline "alternate", ways 3);
MODE=handoff DEPTH=3: the round-3 hand-off pipeline."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import tcresnet_amd as T
from tcresnet_amd.pipeline import InferencePipeline
from bench import synth_batch

dev = torch.device("cuda")
B = 4096
wav = synth_batch(B, dev, 1234)
fe = T.Frontend(window_size_samples=640, window_stride_samples=320, device=dev)
net = T.TCResNet("TCResNet8", [16, 24, 32, 48], 40, fe.n_frames, 12, device=dev)
net.init_xavier(0)
pipe = (InferencePipeline(fe, net, B, depth=int(os.environ.get("DEPTH", "3"))) if os.environ.get("MODE") == "handoff"
        else InferencePipeline(fe, net, B, mode="alternate", ways=int(os.environ.get("WAYS", "3"))))
for _ in range(int(os.environ.get("STEPS", "120"))):
    pipe.submit(wav)
pipe.sync()
torch.cuda.synchronize()
