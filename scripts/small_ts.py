"""Cycles between the barriers of the small-batch network kernel at batch 1 (side library: python scripts/build_whatif.py 2048; WRONG outputs)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch, tcresnet_amd as T
from bench import synth_batch
dev = torch.device("cuda")
lib = T._lib.load_from(os.path.join(ROOT, "tc-resnet_amd", "lib", "whatif", "libtcr_w2048.so"), "hip")
B = int(os.environ.get("B", "1"))
wav = synth_batch(B, dev, 1234)
fe = T.Frontend(window_size_samples=640, window_stride_samples=320, lib=lib, device=dev)
net = T.TCResNet("TCResNet8", [16, 24, 32, 48], 40, fe.n_frames, 12, lib=lib, device=dev); net.init_xavier(0)
feat = fe(wav)
for _ in range(20): lg, pr = net.forward_infer(feat)
torch.cuda.synchronize()
rows = []
for _ in range(10):
    lg, pr = net.forward_infer(feat); torch.cuda.synchronize(); rows.append(pr.cpu().numpy()[0])
v = np.median(np.array(rows), axis=0)
names = ["weights P0+P1 wait", "conv0", "down0+conv0_0", "conv0_1", "down1+conv1_0", "conv1_1", "down2+conv2_0", "conv2_1", "head", "-", "-", "-"]
print("total cycles %.0f = %.1f us at 2.4 GHz" % (v[:9].sum(), v[:9].sum() / 2400))
for i in range(9): print(f"{names[i]:20s} {v[i]:8.0f} cycles")
