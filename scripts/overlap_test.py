import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, tcresnet_amd as T
from bench import synth_batch
dev = torch.device("cuda"); lib = T._lib.get()
B = 4096
wav = synth_batch(B, dev, 1234)
fe = T.Frontend(window_size_samples=640, window_stride_samples=320, device=dev)
net = T.TCResNet("TCResNet8", [16, 24, 32, 48], 40, fe.n_frames, 12, device=dev); net.init_xavier(0)
feats = [torch.empty((B, 40, fe.n_frames + 8), device=dev) for _ in range(2)]
outs = [(torch.empty((B, 12), device=dev), torch.empty((B, 12), device=dev)) for _ in range(2)]

def seq(K):
    for k in range(K):
        fe(wav, out=feats[0]); net.forward_infer(feats[0])

sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
def overlapped(K):
    fe_done = [torch.cuda.Event() for _ in range(K)]
    net_done = [torch.cuda.Event() for _ in range(K)]
    for k in range(K):
        with torch.cuda.stream(sa):
            if k >= 2: sa.wait_event(net_done[k - 2])
            fe(wav, out=feats[k & 1]); fe_done[k].record(sa)
        with torch.cuda.stream(sb):
            sb.wait_event(fe_done[k]); net.forward_infer(feats[k & 1]); net_done[k].record(sb)
    torch.cuda.current_stream().wait_stream(sa); torch.cuda.current_stream().wait_stream(sb)

for fused_off in (0, 1):
    lib.tcr_tune(3, fused_off)
    for name, fn in (("sequential", seq), ("2-stream", overlapped)):
        fn(5); torch.cuda.synchronize()
        t0 = time.perf_counter(); fn(40); torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f"net={('per-layer' if fused_off else 'fused')} {name:10s}: {dt/40*1e3:.3f} ms/step  {B*40/dt/1e6:.2f} M utt/s")
