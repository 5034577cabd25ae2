import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, tcresnet_amd as T
from bench import synth_batch
dev = torch.device("cuda"); lib = T._lib.get()
B = 4096
wav = synth_batch(B, dev, 1234)
def timeit(fn, n=20, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort(); return ts[len(ts)//2]
for tag, win, hop in (("4020", 640, 320), ("3010", 480, 160)):
    fe = T.Frontend(window_size_samples=win, window_stride_samples=hop, device=dev)
    feat = fe(wav)
    for name, scope, ch in (("TCResNet8-1.0", "TCResNet8", [16, 24, 32, 48]), ("TCResNet14-1.5", "TCResNet14", [24, 36, 36, 48, 48, 72, 72])):
        net = T.TCResNet(scope, ch, 40, fe.n_frames, 12, device=dev); net.init_xavier(0)
        lib.tcr_tune(3, 1); ref, _ = net.forward_infer(feat); t_layer = timeit(lambda: net.forward_infer(feat))
        lib.tcr_tune(3, 0)
        res = []
        for nf in (0,):
          lib.tcr_tune(3, nf)
          for w in (0, 404, 408, 416, 808, 816):
            for g in (0, 7, 8):
                lib.tcr_tune(4, g); lib.tcr_tune(5, w)
                out, _ = net.forward_infer(feat); torch.cuda.synchronize()
                err = float((out - ref).abs().max())
                res.append((f"{nf}/{w}/{g}", timeit(lambda: net.forward_infer(feat)), err))
        lib.tcr_tune(4, 0); lib.tcr_tune(5, 0); lib.tcr_tune(3, 0)
        print(f"{name} {tag}: per-layer {t_layer:.1f} us | fused " + " ".join(f"W/G{g}:{t:.1f}us({e:.0e})" for g, t, e in res))
