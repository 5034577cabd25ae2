"""Batch-1 (and small-batch) forward latency: kernel durations under rocprofv3 + host-timed call-to-sync latency.  B=1 STEPS=200"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, tcresnet_amd as T
from bench import synth_batch
dev = torch.device("cuda")
for kv in filter(None, os.environ.get("TUNE", "").split(",")):
    T._lib.get().tcr_tune(int(kv.split("=")[0]), int(kv.split("=")[1]))
B = int(os.environ.get("B", "1"))
wav = synth_batch(B, dev, 1234)
fe = T.Frontend(window_size_samples=640, window_stride_samples=320, device=dev)
net = T.TCResNet("TCResNet8", [16, 24, 32, 48], 40, fe.n_frames, 12, device=dev); net.init_xavier(0)
out = (torch.empty((B, 12), device=dev), torch.empty((B, 12), device=dev))
for _ in range(100): net.forward_waveform(fe, wav, out=out)
torch.cuda.synchronize()
n = int(os.environ.get("STEPS", "200"))
t0 = time.perf_counter()
for _ in range(n):
    net.forward_waveform(fe, wav, out=out); torch.cuda.synchronize()
print(f"B={B}: {(time.perf_counter() - t0) / n * 1e6:.1f} us per call + sync", flush=True)
st = torch.cuda.current_stream(dev)
t0 = time.perf_counter()
for _ in range(n):
    net.forward_waveform(fe, wav, out=out); st.synchronize()
print(f"B={B}: {(time.perf_counter() - t0) / n * 1e6:.1f} us per call + stream synchronize", flush=True)
ev = torch.cuda.Event()
t0 = time.perf_counter()
for _ in range(n):
    net.forward_waveform(fe, wav, out=out); ev.record(st); ev.synchronize()
print(f"B={B}: {(time.perf_counter() - t0) / n * 1e6:.1f} us per call + event record / synchronize", flush=True)
ev2 = torch.cuda.Event(blocking=False)
t0 = time.perf_counter()
for _ in range(n):
    net.forward_waveform(fe, wav, out=out); ev2.record(st)
    while not ev2.query(): pass
print(f"B={B}: {(time.perf_counter() - t0) / n * 1e6:.1f} us per call + event record / busy query", flush=True)
