#!/usr/bin/env python
"""DS-CNN-L eval forwards only (for rocprofv3 runs): N net-only forwards at batch 4096, knob TCR_TUNE_CONV_B from argv[1]."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import tcresnet_amd as T
from bench import synth_batch
dev = torch.device("cuda")
lib = T._lib.get()
lib.tcr_tune(2, int(sys.argv[1]) if len(sys.argv) > 1 else 0)
wav = synth_batch(4096, dev, 1234)
fe3 = T.Frontend(window_size_samples=640, window_stride_samples=320, num_mfccs=10, device=dev)
feat3 = fe3(wav)
ds = T.DSCNN("L", fe3.n_frames, 10, 12, device=dev); ds.init_xavier(0)
for _ in range(6): ds.forward_infer(feat3)
torch.cuda.synchronize()
