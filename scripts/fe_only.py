import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, tcresnet_amd as T
from bench import synth_batch
dev = torch.device("cuda"); lib = T._lib.get()
lib.tcr_tune(1, int(os.environ.get("FE_VAR", "0")))
wav = synth_batch(4096, dev, 1234)
win, hop = (640, 320) if os.environ.get("FE_TAG", "4020") == "4020" else (480, 160)
fe = T.Frontend(window_size_samples=win, window_stride_samples=hop, device=dev)
feat = fe(wav)
for _ in range(int(os.environ.get("FE_N", "6"))): fe(wav, out=feat)
torch.cuda.synchronize()
