#!/usr/bin/env python
"""Phase stamps of the lazy backward kernels (side library scripts/whatif_libs/lib_lazy_ts.so = bwd_lazy.hip compiled -DTCR_LAZY_TS;
TCR_DEBUG_LAZY_TS=1): per launch, three workgroups' clock64 deltas: start | stage barrier, stage, barrier, conv, barrier, epilogue, barrier ..."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["TCR_DEBUG_LAZY_TS"] = "1"
os.environ["TCR_DEBUG_LAZY"] = "1"
import torch
import tcresnet_amd as T
from bench import synth_batch
dev = torch.device("cuda")
lib = T._lib.load_from(os.path.join(ROOT, "scripts", "whatif_libs", "lib_lazy_ts.so"), "hip")
B = 4096
wav = synth_batch(B, dev, 1234)
lab = torch.zeros((B, 12), device=dev); lab[torch.arange(B), torch.arange(B) % 12] = 1
fe = T.Frontend(window_size_samples=640, window_stride_samples=320, lib=lib, device=dev)
feat = fe(wav)
net = T.TCResNet("TCResNet8", [16, 24, 32, 48], 40, fe.n_frames, 12, lib=lib, device=dev); net.init_xavier(0)
for i in range(3):
    if i == 2: print("==== step 2", file=sys.stderr, flush=True)
    net.forward_train(feat, lab, keep_prob=0.5, seed=i); net.backward(); net.sgd_momentum_step(0.1, 0.9, 0.001)
torch.cuda.synchronize()
