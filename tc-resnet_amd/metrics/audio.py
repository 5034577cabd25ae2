"""The reference's non-tensor metric ops for the audio models, evaluated on the stacked softmax outputs / one-hot labels
of an evaluation run (metrics/parser.py:136-147 -> metrics/ops/non_tensor_ops.py; registered by AudioMetricManager,
metrics/manager.py:9-46).  Host-side NumPy / scikit-learn, as in the reference; key names are the reference's
(`accuracy/<split>`, `top5_accuracy/<split>`, `mAP/<split>/<average>`, `precision|recall|f1score|ap/<split>/<label>`,
`classification_report/<split>`)."""
from __future__ import annotations

import warnings
from typing import Dict, Sequence

import numpy as np

_AVERAGES = ("macro", "micro", "weighted", "samples")
_PER_CLASS = ("precision", "recall", "f1score", "ap")


def top_n_accuracy(y_true: np.ndarray, y_pred_onehot: np.ndarray, n: int) -> float:
    """metrics/funcs.py:4-20: the label is among the n highest scores."""
    assert y_true.ndim == 1 and y_pred_onehot.ndim == 2 and y_true.shape[0] == y_pred_onehot.shape[0] and y_pred_onehot.shape[1] >= n
    top = np.argsort(-y_pred_onehot, axis=-1)[:, :n]
    return float(np.mean([label in row for label, row in zip(y_true, top)]))


def audio_metrics(predictions_onehot: np.ndarray, labels_onehot: np.ndarray, label_names: Sequence[str], split: str,
                  use_class_metrics: bool = True) -> Dict[str, object]:
    from sklearn.metrics import (accuracy_score, average_precision_score, classification_report, f1_score, precision_score,
                                 recall_score)
    predictions = predictions_onehot.argmax(axis=-1)            # metrics/parser.py:137-138
    labels = labels_onehot.argmax(axis=-1)
    idx = list(range(len(label_names)))
    out: Dict[str, object] = {}
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")                         # classes absent from a small evaluation set
        out[f"accuracy/{split}"] = float(accuracy_score(labels, predictions))
        if predictions_onehot.shape[1] >= 5:
            out[f"top5_accuracy/{split}"] = top_n_accuracy(labels, predictions_onehot, 5)
        for avg in _AVERAGES:
            out[f"mAP/{split}/{avg}"] = float(average_precision_score(labels_onehot, predictions_onehot, average=avg))
        if use_class_metrics:                                   # registered only below --maximum_num_labels_for_metric labels
            per_class = {"precision": precision_score(labels, predictions, average=None, labels=idx),
                         "recall": recall_score(labels, predictions, average=None, labels=idx),
                         "f1score": f1_score(labels, predictions, average=None, labels=idx),
                         "ap": average_precision_score(labels_onehot, predictions_onehot, average=None)}
            for name, vals in per_class.items():
                for i in idx:
                    out[f"{name}/{split}/{label_names[i]}"] = float(vals[i])
        out[f"classification_report/{split}"] = "[ClassificationReport]\n" + classification_report(labels, predictions, labels=idx,
                                                                                                      target_names=list(label_names))
    return out


def best_keep_metric_with_modes(split: str, label_names: Sequence[str], use_class_metrics: bool, num_classes: int = 12) -> Dict[str, str]:
    """metric key -> "max" / "min" for every metric op flagged is_for_best_keep (MetricManagerBase.get_best_keep_metric_with_modes,
    metrics/base.py:233-239): mAP (4 averages), accuracy, top-5 accuracy, the losses, and the per-class ops when registered."""
    out = {f"mAP/{split}/{avg}": "max" for avg in _AVERAGES}
    out[f"accuracy/{split}"] = "max"
    if max(len(label_names), num_classes) >= 5:
        out[f"top5_accuracy/{split}"] = "max"
    out[f"metric_loss/{split}/total_loss"] = "min"
    out[f"metric_loss/{split}/model_loss"] = "min"
    if use_class_metrics:
        for name in _PER_CLASS:
            for lab in label_names:
                out[f"{name}/{split}/{lab}"] = "max"
    return out
