"""Cycles between the barriers of the static fused eval kernel (TCResNet8-1.0, 49 frames, batch 4096): side library built by
`python scripts/build_whatif.py 2048` (fused.hip with -DTCR_FUSED_WHATIF=2048: WRONG outputs -- the phase times overwrite the
probabilities).  Prints per phase the median over the workgroups and a few individual ones."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch, tcresnet_amd as T
from bench import synth_batch
dev = torch.device("cuda")
lib = T._lib.load_from(os.path.join(ROOT, "tc-resnet_amd", "lib", "whatif", os.environ.get("TS_LIB", "libtcr_w2048.so")), "hip")
wav = synth_batch(4096, dev, 1234)
fe = T.Frontend(window_size_samples=640, window_stride_samples=320, lib=lib, device=dev)
net = T.TCResNet("TCResNet8", [16, 24, 32, 48], 40, fe.n_frames, 12, lib=lib, device=dev); net.init_xavier(0)
feat = fe(wav)
names = ["conv0", "down0+conv0_0", "conv0_1", "down1+conv1_0", "conv1_1", "down2+conv2_0", "conv2_1", "head", "-", "-", "-", "-"]
for knob in (int(x) for x in os.environ.get("TS_KNOBS", "0").split(",")):          # arms of TCR_TUNE_NET_FUSED (8: jobs of two tiles dealt round-robin = rounds 3-5, 9: 16-position units, 0: units + a tap of weight lookahead = round 6)
    lib.tcr_tune(3, knob)
    for _ in range(20): lg, pr = net.forward_infer(feat)
    torch.cuda.synchronize()
    lg, pr = net.forward_infer(feat)
    torch.cuda.synchronize()
    v = pr.cpu().numpy().reshape(-1, 8, 12)[:, 0, :]          # first utterance of every group of 8
    print(f"knob {knob}: groups:", v.shape[0], " total cycles per group (median): %.0f" % np.median(v[:, :8].sum(1)))
    for i in range(8):
        print(f"{names[i]:16s} median {np.median(v[:, i]):8.0f}  p10 {np.percentile(v[:, i], 10):8.0f}  p90 {np.percentile(v[:, i], 90):8.0f} cycles")
lib.tcr_tune(3, 0)
