"""Evaluation metrics with the reference's keys (metrics/ops/non_tensor_ops.py, metrics/funcs.py, metrics/parser.py)."""
from .audio import audio_metrics, best_keep_metric_with_modes, top_n_accuracy  # noqa: F401
