"""Does splitting the next batch's front-end between the forward and the backward beat issuing it whole in front of the forward?  (TCResNet8,
batch 4096: whole-before 1200 us, halves 1239, quarter / three quarters before 1243 / 1223, whole-after 1225: no.)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, tcresnet_amd as T
from bench import synth_batch
dev = torch.device("cuda")
B = 4096
wav = synth_batch(B, dev, 1234)
lab = torch.zeros((B, 12), device=dev); lab[torch.arange(B), torch.arange(B) % 12] = 1
def timeit(fn, n=40, warm=10):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
fe = T.Frontend(window_size_samples=640, window_stride_samples=320, device=dev)
net = T.TCResNet("TCResNet8", [16, 24, 32, 48], 40, fe.n_frames, 12, device=dev); net.init_xavier(0)
s2 = torch.cuda.Stream(dev)
bufs = [torch.empty((B, 40, fe.n_frames + 8), device=dev) for _ in range(2)]
fe(wav, out=bufs[0]); fe(wav, out=bufs[1]); torch.cuda.synchronize()
for mode in ("before", "split_half", "split_quarter_before", "split_3q_before", "after"):
    k = [0]
    ready = [None, None]
    def sub(lo, hi, i):
        cur = torch.cuda.current_stream(dev)
        with torch.cuda.stream(s2):
            s2.wait_stream(cur)
            fe(wav[lo:hi], out=bufs[i][lo:hi])
            ev = torch.cuda.Event(); ev.record(s2); ready[i] = ev
    def step():
        i = k[0] % 2; j = 1 - i; k[0] += 1
        cur = torch.cuda.current_stream(dev)
        if ready[i] is not None: cur.wait_event(ready[i])
        f = bufs[i]
        cut = {"before": B, "split_half": B // 2, "split_quarter_before": B // 4, "split_3q_before": 3 * B // 4, "after": 0}[mode]
        if cut > 0: sub(0, cut, j)
        net.forward_train(f, lab, keep_prob=0.5, seed=1)
        if cut < B: sub(cut, B, j)
        net.backward(); net.sgd_momentum_step(0.1, 0.9, 0.001)
    print(f"{mode:22s}: {timeit(step):8.1f} us", flush=True)
