"""`python -m tcresnet_amd.train_audio <global flags> <ModelName> <model flags>` -- same command-line shape as
the reference's train_audio.py (:19-77).  Data: --dataset_path <dir of WAVs> or `synthetic` (the tf.data file handling is outside
the hot path).  Multi-GPU: launch with torch.distributed.run, one process per GPU (RCCL; TCR_DIST_BACKEND=gloo for a CPU-side
rehearsal of the collective calls)."""
from __future__ import annotations

import argparse
import logging
import os
from typing import List

import torch
import torch.distributed as dist

from .datasets.audio_data_wrapper import SingleLabelAudioDataWrapper
from .datasets.synthetic import SyntheticAudioDataWrapper
from .factory import audio_nets
from .factory.base import TFModel
from .helper.base import Base
from .helper.trainer import SingleLabelAudioTrainer, TrainerBase


def add_data_arguments(parser):
    g = parser.add_argument_group("(Data) Arguments")       # datasets/data_wrapper_base.py:250-288, audio_data_wrapper.py:61-110
    g.add_argument("--dataset_path", default="synthetic", type=str)
    g.add_argument("--dataset_split_name", default=["train"], type=str, nargs="*")
    g.add_argument("--batch_size", default=32, type=int)
    g.add_argument("--augmentation_method", default="no_augmentation_audio", type=str)
    g.add_argument("--num_threads", default=8, type=int)
    g.add_argument("--sample_rate", default=16000, type=int)
    g.add_argument("--clip_duration_ms", default=1000, type=int)
    g.add_argument("--window_size_ms", default=30.0, type=float)
    g.add_argument("--window_stride_ms", default=10.0, type=float)
    g.add_argument("--lower_edge_hertz", default=80.0, type=float)
    g.add_argument("--upper_edge_hertz", default=7600.0, type=float)
    g.add_argument("--num_mel_bins", default=64, type=int)
    g.add_argument("--num_mfccs", default=40, type=int)
    g.add_argument("--num_silent", default=-1, type=int)
    g.add_argument("--background_max_volume", default=0.1, type=float)
    g.add_argument("--background_frequency", default=0.8, type=float)
    g.add_argument("--shuffle", dest="shuffle", action="store_true")
    g.add_argument("--no-shuffle", dest="shuffle", action="store_false")
    g.set_defaults(shuffle=True)
    g.add_argument("--add_null_class", dest="add_null_class", action="store_true")
    g.add_argument("--no-add_null_class", dest="add_null_class", action="store_false")
    g.set_defaults(add_null_class=True)
    g.add_argument("--buffer_size", default=1000, type=int)
    g.add_argument("--prefetch_factor", default=100, type=int)


def add_metric_arguments(parser):
    g = parser.add_argument_group("Metric Manager Arguments")       # metrics/base.py:249-260
    g.add_argument("--exclude_metric_names", nargs="*", default=[], type=str)
    g.add_argument("--max_summary_outputs", default=3, type=int)


def add_model_subparsers(parser):
    subparsers = parser.add_subparsers(title="Model", description="")
    for class_name in audio_nets._available_nets:
        sub = subparsers.add_parser(class_name)
        sub.add_argument("--model", default=class_name, type=str, help="DO NOT FIX ME")
        getattr(audio_nets, class_name).add_arguments(sub)


def parse_arguments(arguments: List[str] = None):
    parser = argparse.ArgumentParser(description=__doc__)
    add_model_subparsers(parser)
    TFModel.add_arguments(parser)
    audio_nets.AudioNetModel.add_arguments(parser)
    add_data_arguments(parser)
    Base.add_arguments(parser)
    TrainerBase.add_arguments(parser)
    SingleLabelAudioTrainer.add_arguments(parser)
    add_metric_arguments(parser)
    return parser.parse_args(arguments)


def init_distributed():
    if "RANK" in os.environ and int(os.environ.get("WORLD_SIZE", "1")) > 1 and not dist.is_initialized():
        backend = os.environ.get("TCR_DIST_BACKEND", "nccl")
        if torch.cuda.is_available():
            torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")) % torch.cuda.device_count())
        dist.init_process_group(backend)


def train(args):
    logging.basicConfig(level=logging.INFO)
    init_distributed()
    is_training = True
    dataset_name = args.dataset_split_name[0]
    wrapper = SyntheticAudioDataWrapper if args.dataset_path == "synthetic" else SingleLabelAudioDataWrapper
    dataset = wrapper(args, None, dataset_name, is_training)
    wavs, labels = dataset.get_input_and_output_op()
    model = getattr(audio_nets, args.model)(args, dataset)
    model.build(wavs=wavs, labels=labels, is_training=is_training)
    trainer = SingleLabelAudioTrainer(model, None, args, dataset, dataset_name)
    trainer.train()
    return trainer


if __name__ == "__main__":
    train(parse_arguments())
