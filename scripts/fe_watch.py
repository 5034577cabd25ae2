"""Front-end / network kernel time next to the clocks and socket power sampled WHILE the kernel loops (a box-state check: on 2026-10-01 one box ran the
front-end at 199 us instead of 158 with the network kernel unchanged)."""
import os, sys, time, subprocess, threading
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import torch, tcresnet_amd as T
from bench import synth_batch
dev = torch.device("cuda")
wav = synth_batch(4096, dev, 1234)
fe = T.Frontend(window_size_samples=640, window_stride_samples=320, device=dev)
net = T.TCResNet("TCResNet8", [16, 24, 32, 48], 40, fe.n_frames, 12, device=dev); net.init_xavier(0)
out = fe(wav).clone()
o2 = (torch.empty((4096, 12), device=dev), torch.empty((4096, 12), device=dev))
samples, stop = [], [False]
def sampler():
    while not stop[0]:
        r = subprocess.run("rocm-smi --showclocks --showpower 2>/dev/null | grep -i 'sclk\\|mclk\\|Power (W)'", shell=True, capture_output=True, text=True).stdout
        samples.append(" ".join(x.split(":")[-1].strip() for x in r.strip().splitlines()))
def loop(fn, secs):
    samples.clear(); stop[0] = False
    th = threading.Thread(target=sampler); th.start()
    n = 0; torch.cuda.synchronize(); t0 = time.perf_counter()
    while time.perf_counter() - t0 < secs:
        for _ in range(100): fn()
        torch.cuda.synchronize(); n += 100
    dt = time.perf_counter() - t0
    stop[0] = True; th.join()
    return dt / n * 1e6, list(samples)
for name, fn in (("front-end", lambda: fe(wav, out=out)), ("network", lambda: net.forward_infer(out, out=o2)), ("front-end", lambda: fe(wav, out=out))):
    us, s = loop(fn, 4.0)
    print(f"{name}: {us:.1f} us per launch; sampled while running: {s[1:6]}", flush=True)
