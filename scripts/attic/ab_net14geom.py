import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import tcresnet_amd as T
from bench import synth_batch
dev = torch.device("cuda"); lib = T._lib.get(); B = 4096
wav = synth_batch(B, dev, 1234)
def wall(fn, n=60, warm=20):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6
for tag, win, hop in (("4020", 640, 320), ("3010", 480, 160)):
    fe = T.Frontend(window_size_samples=win, window_stride_samples=hop, device=dev)
    net = T.TCResNet("TCResNet14", [24, 36, 36, 48, 48, 72, 72], 40, fe.n_frames, 12, device=dev); net.init_xavier(0)
    feat = fe(wav)
    for g in (0, 3, 4, 5, 8, 10):
        for wv in (0, 8, 16):
            lib.tcr_tune(4, g); lib.tcr_tune(5, wv)
            try: print(f"{tag} TCResNet14-1.5 group {g} waves {wv}: {wall(lambda: net.forward_infer(feat)):7.1f} us", flush=True)
            except Exception as e: print(g, wv, "failed", str(e)[:80])
    lib.tcr_tune(4, 0); lib.tcr_tune(5, 0)
