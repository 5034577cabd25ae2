"""`python -m tcresnet_amd.freeze <flags> <ModelName> <model flags>` (reference freeze.py:16-83): build the deployable model
WITHOUT preprocessing on a [1, height, width, channels] input, restore --checkpoint_path (scope filters / --use_ema as in
the evaluator), convert the variables to constants and write `<checkpoint>.pb` next to the checkpoint.

The file is not a TensorFlow GraphDef: it is the frozen plan of the MI355X kernels (tcresnet_amd/deploy.py) -- conv / fc
weights under their TF names, BatchNorm folded to (scale, shift) -- which `deploy.FrozenModel.load` runs directly."""
from __future__ import annotations

import argparse
import logging
from pathlib import Path
from typing import List

import torch

from . import runtime
from .common.model_loader import Ckpt
from .factory import audio_nets
from .factory.base import TFModel
from .helper.base import Base
from .train_audio import add_model_subparsers


def parse_arguments(arguments: List[str] = None):
    parser = argparse.ArgumentParser(description=__doc__)
    add_model_subparsers(parser)
    TFModel.add_arguments(parser)
    audio_nets.AudioNetModel.add_arguments(parser)
    Base.add_arguments(parser)
    parser.add_argument("--width", required=True, type=int)
    parser.add_argument("--height", required=True, type=int)
    parser.add_argument("--channels", required=True, type=int)
    parser.add_argument("--sample_rate", type=int, default=16000)
    parser.add_argument("--clip_duration_ms", type=int)
    parser.add_argument("--window_size_ms", type=float, default=30.0)
    parser.add_argument("--window_stride_ms", type=float, default=30.0)
    parser.add_argument("--num_mel_bins", type=int, default=64)
    parser.add_argument("--num_mfccs", type=int, default=64)
    parser.add_argument("--lower_edge_hertz", type=float, default=80.0)
    parser.add_argument("--upper_edge_hertz", type=float, default=7600.0)
    return parser.parse_args(arguments)


def freeze(args) -> str:
    logging.basicConfig(level=logging.INFO)
    model = getattr(audio_nets, args.model)(args)
    input_tensors, output_tensor = model.build_deployable_model(include_preprocess=False)
    Ckpt(model.engine, include_scopes=args.checkpoint_include_scopes, exclude_scopes=args.checkpoint_exclude_scopes,
         ignore_missing_vars=args.ignore_missing_vars, use_ema=args.use_ema, ema_decay=args.ema_decay).load(args.checkpoint_path)
    frozen = model.freeze()
    checkpoint_path = Path(args.checkpoint_path)
    out = checkpoint_path.parent / f"{checkpoint_path.name}.pb"
    frozen.save(str(out))
    print(f"Save freezed pb : {out}")
    return str(out)


if __name__ == "__main__":
    freeze(parse_arguments())
