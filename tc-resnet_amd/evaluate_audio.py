"""`python -m tcresnet_amd.evaluate_audio <global flags> <ModelName> <model flags>` (reference evaluate_audio.py:19-87):
is_training=False build (BN moving statistics, no dropout), evaluate a checkpoint once."""
from __future__ import annotations

import argparse
import logging
from typing import List

from .datasets.audio_data_wrapper import SingleLabelAudioDataWrapper
from .datasets.synthetic import SyntheticAudioDataWrapper
from .factory import audio_nets
from .factory.base import TFModel
from .helper.evaluator import SingleLabelAudioEvaluator
from .train_audio import add_data_arguments


def parse_arguments(arguments: List[str] = None):
    parser = argparse.ArgumentParser(description=__doc__)
    subparsers = parser.add_subparsers(title="Model", description="")
    SingleLabelAudioEvaluator.add_arguments(parser)
    add_data_arguments(parser)
    TFModel.add_arguments(parser)
    audio_nets.AudioNetModel.add_arguments(parser)
    for class_name in audio_nets._available_nets:
        sub = subparsers.add_parser(class_name)
        sub.add_argument("--model", default=class_name, type=str, help="DO NOT FIX ME")
        getattr(audio_nets, class_name).add_arguments(sub)
    return parser.parse_args(arguments)


def main(args):
    logging.basicConfig(level=logging.INFO)
    wrapper = SyntheticAudioDataWrapper if args.dataset_path == "synthetic" else SingleLabelAudioDataWrapper
    dataset = wrapper(args, None, args.dataset_split_name[0], False)
    wavs, labels = dataset.get_input_and_output_op()
    model = getattr(audio_nets, args.model)(args, dataset)
    model.build(wavs=wavs, labels=labels, is_training=False)
    evaluator = SingleLabelAudioEvaluator(model, None, args, dataset, args.dataset_split_name[0])
    return evaluator.evaluate_once(args.checkpoint_path or None)


if __name__ == "__main__":
    main(parse_arguments())
