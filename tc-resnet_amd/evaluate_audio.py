"""`python -m tcresnet_amd.evaluate_audio <global flags> <ModelName> <model flags>` (reference evaluate_audio.py:19-87):
is_training=False build (BN moving statistics, no dropout); `--valid_type once` evaluates --checkpoint_path (a checkpoint
prefix, or a directory -> its latest checkpoint), `--valid_type loop` watches the directory and evaluates every new
checkpoint until one at or beyond --max_step_from_restore has been seen."""
from __future__ import annotations

import argparse
import logging
from typing import List

from .common.tf_utils import ckpt_iterator
from .datasets.audio_data_wrapper import SingleLabelAudioDataWrapper
from .datasets.synthetic import SyntheticAudioDataWrapper
from .factory import audio_nets
from .factory.base import TFModel
from .helper.base import Base
from .helper.evaluator import Evaluator, SingleLabelAudioEvaluator
from .train_audio import add_data_arguments, add_metric_arguments, add_model_subparsers


def parse_arguments(arguments: List[str] = None):
    parser = argparse.ArgumentParser(description=__doc__)
    add_model_subparsers(parser)
    Base.add_arguments(parser)
    Evaluator.add_arguments(parser)
    add_data_arguments(parser)
    TFModel.add_arguments(parser)
    audio_nets.AudioNetModel.add_arguments(parser)
    add_metric_arguments(parser)
    return parser.parse_args(arguments)


def main(args):
    logging.basicConfig(level=logging.INFO)
    log = logging.getLogger("EvaluateAudio")
    is_training = False
    dataset_name = args.dataset_split_name[0]
    wrapper = SyntheticAudioDataWrapper if args.dataset_path == "synthetic" else SingleLabelAudioDataWrapper
    dataset = wrapper(args, None, dataset_name, is_training)
    wavs, labels = dataset.get_input_and_output_op()
    model = getattr(audio_nets, args.model)(args, dataset)
    model.build(wavs=wavs, labels=labels, is_training=is_training)
    evaluator = SingleLabelAudioEvaluator(model, None, args, dataset, dataset_name)
    if args.valid_type == "once":
        return evaluator.evaluate_once(args.checkpoint_path)
    if args.valid_type == "loop":
        log.info("Start Loop: watching %s", evaluator.watch_path)
        results = []
        for ckpt_path in ckpt_iterator(evaluator.watch_path, min_interval_secs=0, timeout=getattr(args, "loop_timeout_secs", None), logger=log):
            log.info("[watch] %s", ckpt_path)
            results.append(evaluator.evaluate_once(ckpt_path))
            if evaluator.finished:          # the reference calls sys.exit() here (helper/evaluator.py:131-133)
                break
        return results
    raise ValueError(f"Undefined valid_type: {args.valid_type}")


if __name__ == "__main__":
    main(parse_arguments())
