"""TEST INFRASTRUCTURE ONLY -- CPU restatement (oracle) of the TC-ResNet hot path.

PARITY UNPINNED: the reference (hyperconnect/TC-ResNet) is TensorFlow-1.13.1
graph code with no tests, fixtures, golden vectors or checkpoints, and TF 1.13.1
cannot be installed in this image (no wheel, Python 3.10).  This package restates
the TF op semantics from the reference call sites (cited per function) in
NumPy float64; `torch_ref.py` is an independent PyTorch-CPU float32 second
implementation used to cross-check it and as the timed CPU baseline.

Nothing under `oracle/` may be imported by the product package
(`tc-resnet_amd/`).  Only `tests/`, `__graft_entry__.smoke()` and
`bench.py`'s `cpu_baseline` leg use it, and only as the checker.
"""
