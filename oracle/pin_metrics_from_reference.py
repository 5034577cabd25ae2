"""TEST INFRASTRUCTURE ONLY -- pins the ONE piece of the reference that runs in this image without TensorFlow: metrics/funcs.py
(`topN_accuracy`, pure NumPy; the top-5 accuracy of `metrics/ops/non_tensor_ops.py`, reported by evaluate_audio.py).  Everything else
under /root/reference imports tensorflow 1.13 (or humanfriendly / overload, also absent) at module level.

    python oracle/pin_metrics_from_reference.py            # (in the build container: /root/reference must exist)

Imports the reference's module, runs it on seeded inputs (ordinary scores, ties, a label never in the top N, N = number of classes) and
stores inputs + the reference's outputs as tests/golden/metrics_topn.npz.  tests/test_oracle.py::test_top_n_accuracy_pinned_to_the_reference
checks tcresnet_amd.metrics.top_n_accuracy against these vectors on any box; nothing here travels to the GPU box except the fixture."""
import os
import sys

import numpy as np

REF = os.environ.get("TCR_REFERENCE", "/root/reference")


def main():
    if not os.path.isdir(os.path.join(REF, "metrics")):
        sys.exit(f"{REF}/metrics not found: this script runs where the reference checkout is")
    sys.path.insert(0, REF)
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_metrics_funcs", os.path.join(REF, "metrics", "funcs.py"))   # (metrics/__init__ is empty; ops/ needs TF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    rng = np.random.RandomState(2024)
    out = {}
    cases = []
    for i, (n_samples, n_classes) in enumerate(((64, 12), (257, 12), (39, 6), (100, 12))):
        y_true = rng.randint(0, n_classes, n_samples).astype(np.int64)
        scores = rng.rand(n_samples, n_classes).astype(np.float32)
        if i == 2:
            scores = np.round(scores * 4) / 4                  # many exact ties
        if i == 3:
            scores[np.arange(n_samples), y_true] = -1.0        # the label is always last
        out[f"y_true_{i}"], out[f"scores_{i}"] = y_true, scores
        for n in (1, 3, 5, n_classes):
            out[f"acc_{i}_{n}"] = np.float64(mod.topN_accuracy(y_true, scores, n))
        cases.append(i)
    out["cases"] = np.array(cases)
    dst = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "metrics_topn.npz")
    np.savez_compressed(dst, **out)
    print("wrote", dst, {k: float(v) for k, v in out.items() if k.startswith("acc_")})


if __name__ == "__main__":
    main()
