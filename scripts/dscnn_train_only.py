"""A few DS-CNN-L training steps on precomputed features (profiling target; SIZE=S|M|L, STEPS, TUNE=knob=value,...)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, tcresnet_amd as T
from bench import synth_batch
dev = torch.device("cuda")
for kv in filter(None, os.environ.get("TUNE", "").split(",")):
    T._lib.get().tcr_tune(int(kv.split("=")[0]), int(kv.split("=")[1]))
B = int(os.environ.get("BATCH", "4096"))
wav = synth_batch(B, dev, 1234)
lab = torch.zeros((B, 12), device=dev); lab[torch.arange(B), torch.arange(B) % 12] = 1
fe = T.Frontend(window_size_samples=640, window_stride_samples=320, num_mfccs=10, device=dev)
feat = fe(wav)
ds = T.DSCNN(os.environ.get("SIZE", "L"), fe.n_frames, 10, 12, device=dev); ds.init_xavier(0)
for i in range(int(os.environ.get("STEPS", "4"))):
    ds.forward_train(feat, lab); ds.backward(); ds.adam_step(5e-4, i + 1)
torch.cuda.synchronize()
