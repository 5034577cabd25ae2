#!/usr/bin/env python
"""DS-CNN-L eval forward (net only, batch 4096): whole batches on one stream against whole batches alternating between WAYS streams
(one DSCNN instance = one set of activation buffers per stream, same parameters)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import tcresnet_amd as T
from bench import synth_batch
dev = torch.device("cuda")
B = 4096
wav = synth_batch(B, dev, 1234)
fe = T.Frontend(window_size_samples=640, window_stride_samples=320, num_mfccs=10, device=dev)
feat = fe(wav)
for ways in (1, 2, 3):
    nets = []
    for i in range(ways):
        ds = T.DSCNN("L", fe.n_frames, 10, 12, device=dev); ds.init_xavier(0); nets.append(ds)
    streams = [torch.cuda.Stream() for _ in range(ways)]
    def run(n):
        for i in range(n):
            with torch.cuda.stream(streams[i % ways]):
                nets[i % ways].forward_infer(feat)
    run(6); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    cur = torch.cuda.current_stream()
    n = 30
    e0.record()
    for s in streams: s.wait_stream(cur)
    run(n)
    for s in streams: cur.wait_stream(s)
    e1.record(); e1.synchronize()
    print(f"ways {ways}: {e0.elapsed_time(e1) * 1e3 / n:8.1f} us per batch", flush=True)
    del nets
