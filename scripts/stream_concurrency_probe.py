#!/usr/bin/env python
"""Which streams actually run concurrently?  A long spin on stream A, a tiny op on stream B: B finishing first means the two are on
different hardware queues.  Prints the matrix for the null stream + N fresh streams, then times the two-stream inference pipeline on a
concurrent and on a serialised pair."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import tcresnet_amd as T
from tcresnet_amd import pipeline as P
from bench import synth_batch

dev = torch.device("cuda")
B = 4096
wav = synth_batch(B, dev, 1234)
lab = torch.zeros((B, 12), device=dev); lab[torch.arange(B), torch.arange(B) % 12] = 1
fe = T.Frontend(window_size_samples=640, window_stride_samples=320, device=dev)
net = T.TCResNet("TCResNet8", [16, 24, 32, 48], 40, fe.n_frames, 12, device=dev)
net.init_xavier(0)
feat = fe(wav)
if os.environ.get("LIB_FIRST", "1") == "1":      # the library's internal streams exist before ours (as in bench.py)
    net.forward_train(feat, lab, keep_prob=0.5, seed=1); net.backward()
torch.cuda.synchronize()
tiny = torch.zeros(64, device=dev)


def concurrent(a, b, cycles=20_000_000):
    ea, eb = torch.cuda.Event(), torch.cuda.Event()
    torch.cuda.synchronize()
    with torch.cuda.stream(a):
        torch.cuda._sleep(cycles); ea.record(a)
    with torch.cuda.stream(b):
        tiny.add_(1.0); eb.record(b)
    eb.synchronize()
    c = not ea.query()
    torch.cuda.synchronize()
    return c


lib = T._lib.get()
libs = [torch.cuda.ExternalStream(lib.tcr_internal_stream(i, torch.cuda.current_stream(dev).cuda_stream), device=dev) for i in range(4)]
streams = [torch.cuda.default_stream(dev)] + libs + [torch.cuda.Stream(dev) for _ in range(int(os.environ.get("NS", "4")))]
print("streams: 0 = null, 1-4 = the library's internal 0-3, 5.. = fresh")
n = len(streams)
M = [[(i != j and concurrent(streams[i], streams[j])) for j in range(n)] for i in range(n)]
for i in range(n):
    print(f"stream {i}: " + " ".join("C" if M[i][j] else "." for j in range(n)), flush=True)


def wall(fn, n=150, warm=50):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


out = (torch.empty((B, 12), device=dev), torch.empty((B, 12), device=dev))
def seq():
    fe(wav, out=feat); net.forward_infer(feat, out=out)
print(f"sequential: {wall(seq):7.1f} us")
pipe = P.InferencePipeline(fe, net, B, depth=3)
print(f"pipeline on pipeline.shared_stream's streams: {wall(lambda: pipe.submit(wav)):7.1f} us; 50 steps after a drain: {wall(lambda: pipe.submit(wav), n=50, warm=0):7.1f}")
pipe.sync(); torch.cuda.synchronize()
for i in range(1, n):
    for j in range(i + 1, n):
        if (i, j) not in ((3, 4), (5, 6), (5, 7), (6, 7), (3, 5)): continue
        P._STREAMS.clear()
        P._STREAMS[("cuda", torch.cuda.current_device(), "frontend")] = streams[i]
        P._STREAMS[("cuda", torch.cuda.current_device(), "network")] = streams[j]
        pipe = P.InferencePipeline(fe, net, B, depth=3)
        t = wall(lambda: pipe.submit(wav))
        pipe.sync(); torch.cuda.synchronize()
        print(f"pipeline on streams ({i}, {j}) concurrent={M[i][j] and M[j][i]} vs null: fe {M[0][i]} net {M[0][j]}: {t:7.1f} us", flush=True)
