"""The deployable ("frozen") model: what the reference gets from build_deployable_model + freeze.py
(factory/audio_nets.py:87-125, freeze.py:16-49: a graph whose variables were converted to constants, fed either the raw
waveform `input/audio/before_preprocessing` [input_batch_size, samples, 1] through the DEPLOY-path MFCC, or features
`input` [1, height, width, channels]).

Here the artifact is the plan the fused eval kernel consumes directly -- no graph interpreter in between:
  * a JSON description (model name / scope / channels / input + output node names / front-end settings),
  * the constants: conv / fc weights under their TF variable names and, for TC-ResNet, every BatchNorm folded to a
    per-channel (scale, shift) table (tcr_net_fold_bn); DS-CNN keeps its variables and folds inside its eval kernel,
written as one `.npz` next to the checkpoint.  `FrozenModel.load(path)(x)` runs it on the HIP kernels.
"""
from __future__ import annotations

import json
from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

from . import runtime
from ._lib import HALO
from .engine import DSCNN, Frontend, Graph2D, TCResNet, features_to_planar

FORMAT = "tcresnet_amd.frozen/1"


@dataclass
class TensorSpec:
    """Stands in for the tf.placeholder / output tensor the reference's build_deployable_model returns."""
    name: str
    shape: Tuple[int, ...]
    dtype: str = "float32"

    @property
    def op(self):               # `output_tensor.op.name` is what freeze.py passes to convert_variables_to_constants
        return self


class FrozenModel:
    def __init__(self, meta: Dict, constants: Dict[str, np.ndarray], lib=None, device=None):
        if meta.get("format") != FORMAT:
            raise ValueError(f"not a {FORMAT} artifact")
        self.meta, self.constants = meta, constants
        lib = lib if lib is not None else runtime.default_lib()
        device = device if device is not None else runtime.default_device()
        fam = meta["family"]
        if fam == "tcresnet":
            self.engine = TCResNet(meta["scope"], meta["channels"], meta["width"], meta["height"], meta["num_classes"],
                                   bn_decay=meta["bn_decay"], bn_eps=meta["bn_eps"], lib=lib, device=device)
            self.engine.load_state_dict({k: v for k, v in constants.items() if k.endswith("/weights")}, strict=False)
            self.frozen_ss = torch.from_numpy(np.ascontiguousarray(constants["__folded_batch_norm__"])).to(self.engine.device)
        elif fam == "dscnn":
            self.engine = DSCNN(meta["size"], meta["height"], meta["width"], meta["num_classes"], lib=lib, device=device)
            self.engine.load_state_dict(constants)
            self.frozen_ss = None
        elif fam == "graph2d":
            import argparse
            from .factory import audio_nets
            saved = runtime.default_lib(), runtime.default_device()
            runtime.set_default(lib, device)
            try:
                model = getattr(audio_nets, meta["model"])(argparse.Namespace(**meta["args"]))
                self.engine = model._engine_for(torch.empty((1, meta["height"], meta["width"], 1)))
            finally:
                runtime.set_default(*saved)
            self.engine.load_state_dict(constants)
            self.frozen_ss = None
        else:
            raise ValueError(f"unknown model family {fam}")
        self.frontend: Optional[Frontend] = None
        if meta["include_preprocess"]:
            fe = meta["frontend"]
            self.frontend = Frontend(sample_rate=fe["sample_rate"], clip_duration_ms=fe["clip_duration_ms"],
                                     window_size_samples=fe["window_size_samples"], window_stride_samples=fe["window_stride_samples"],
                                     num_mel_bins=fe["num_mel_bins"], num_mfccs=fe["num_mfccs"], lower_edge_hertz=fe["lower_edge_hertz"],
                                     upper_edge_hertz=fe["upper_edge_hertz"], method=fe["method"], lib=lib, device=device)

    @property
    def input_tensors(self) -> List[TensorSpec]:
        return [TensorSpec(i["name"], tuple(i["shape"])) for i in self.meta["inputs"]]

    @property
    def output_tensor(self) -> TensorSpec:
        return TensorSpec(self.meta["output"]["name"], tuple(self.meta["output"]["shape"]))

    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        """x: waveforms [B, samples, 1] (include_preprocess) or features [B, height, width, channels]; returns softmax [B, classes]
        (the node named --output_name).  Any batch size runs; the reference's graph is fixed to input_batch_size."""
        if self.frontend is not None:
            planar = self.frontend(x if x.is_contiguous() else x.contiguous())
        else:
            planar = features_to_planar(x, lib=self.engine.lib)
        if self.frozen_ss is not None:
            return self.engine.forward_frozen(planar, self.frozen_ss)[1]
        return self.engine.forward_infer(planar)[1]

    # ---- file format ----------------------------------------------------------------------------------------------
    def save(self, path: str) -> str:
        out = {"__meta__": np.frombuffer(json.dumps(self.meta, sort_keys=True).encode(), dtype=np.uint8)}
        out.update(self.constants)
        with open(path, "wb") as fh:
            np.savez(fh, **out)
        return path

    @classmethod
    def load(cls, path: str, lib=None, device=None) -> "FrozenModel":
        with np.load(path, allow_pickle=False) as z:
            meta = json.loads(bytes(z["__meta__"]).decode())
            consts = {k: z[k] for k in z.files if k != "__meta__"}
        return cls(meta, consts, lib=lib, device=device)


def export_frozen(model, include_preprocess: bool, inputs: List[TensorSpec], output: TensorSpec) -> FrozenModel:
    """Variables -> constants for the model's current weights (graph_util.convert_variables_to_constants)."""
    eng, args = model.engine, model.args
    sd = eng.state_dict()
    meta = {"format": FORMAT, "model": type(model).__name__, "num_classes": int(args.num_classes), "include_preprocess": bool(include_preprocess),
            "inputs": [{"name": i.name, "shape": list(i.shape)} for i in inputs], "output": {"name": output.name, "shape": list(output.shape)},
            "height": int(args.height), "width": int(args.width), "channels": int(args.channels)}
    if isinstance(eng, TCResNet):
        meta.update(family="tcresnet", scope=eng.scope, channels=[int(c) for c in eng.channels], bn_decay=float(eng.cfg.bn_decay),
                    bn_eps=float(eng.cfg.bn_eps))
        consts = {k: v for k, v in sd.items() if k.endswith("/weights")}
        consts["__folded_batch_norm__"] = eng.fold_bn().cpu().numpy()
    elif isinstance(eng, DSCNN):
        meta.update(family="dscnn", size=eng.size)
        consts = dict(sd)
    elif isinstance(eng, Graph2D):
        meta.update(family="graph2d", args={k: v for k, v in vars(args).items() if isinstance(v, (int, float, str, bool)) or v is None})
        consts = dict(sd)
    else:
        raise NotImplementedError(f"frozen export of {type(eng).__name__}")
    if include_preprocess:
        fe = model._preprocessor._frontend
        meta["frontend"] = {"sample_rate": int(fe.cfg.sample_rate), "clip_duration_ms": int(fe.cfg.n_samples * 1000 // fe.cfg.sample_rate),
                            "window_size_samples": int(fe.cfg.win), "window_stride_samples": int(fe.cfg.hop), "num_mel_bins": int(fe.cfg.n_mel),
                            "num_mfccs": int(fe.cfg.n_coef), "lower_edge_hertz": float(fe.cfg.lower_hz), "upper_edge_hertz": float(fe.cfg.upper_hz),
                            "method": {0: "mfcc", 1: "log_mel_spectrogram", 2: "mfcc_deploy"}[int(fe.cfg.method)]}
    return FrozenModel(meta, consts, lib=eng.lib, device=eng.device)
