"""The headline workload as a ONE-STREAM sequence (front-end, network back to back; batch 4096, 49x40 MFCC): the profiling target whose
rocprofv3 kernel-trace average is the solo duration of each kernel -- what bench.py's roofline.kernel_ms is measured against (in the
bench's timed region the two kernels co-run on two streams).  STEPS (default 300) launches after 100 pre-warm launches."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, tcresnet_amd as T
from bench import synth_batch
dev = torch.device("cuda")
B = 4096
wav = synth_batch(B, dev, 1234)
fe = T.Frontend(window_size_samples=640, window_stride_samples=320, device=dev)
net = T.TCResNet("TCResNet8", [16, 24, 32, 48], 40, fe.n_frames, 12, device=dev); net.init_xavier(0)
feat = torch.empty((B, 40, fe.n_frames + 8), device=dev)
out = (torch.empty((B, 12), device=dev), torch.empty((B, 12), device=dev))
for _ in range(100 + int(os.environ.get("STEPS", "300"))):
    fe(wav, out=feat); net.forward_infer(feat, out=out)
torch.cuda.synchronize()
