"""Front-end kernels A/B in one process: the three-waves-per-SIMD kernel (frontend_pk3.hip, knob 23 = 0) against the two-waves kernel
(frontend_pk.hip, knob 23 = 1), batch 4096, HIP events over interleaved blocks of launches; optional sweep of the rounds per chunk
(knob 1 = 10 + rounds) and of the grid cap (knob 13).  Prints microseconds per launch and whether the features are bitwise equal.

    python scripts/ab_fe_kernel.py [--sweep] [--batch 4096]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tcresnet_amd as T
from bench import synth_batch


def time_launches(fe, wav, out, n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fe(wav, out=out)
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) * 1000.0 / n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sweep", action="store_true")
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--n", type=int, default=40)
    args = ap.parse_args()
    dev = torch.device("cuda")
    wav = synth_batch(args.batch, dev, 1234)
    for win, hop, method in ((640, 320, "mfcc"), (480, 160, "mfcc"), (640, 320, "log_mel_spectrogram")):
        fe = T.Frontend(window_size_samples=win, window_stride_samples=hop, method=method, device=dev)
        tune = fe.lib.tcr_tune
        outs = {}
        for k in (0, 1):
            tune(23, k)
            outs[k] = fe(wav).clone()
        tune(23, 0)
        out = torch.empty_like(outs[0])
        for k in (0, 1):
            tune(23, k)
            time_launches(fe, wav, out, 10)
        best = {0: [], 1: []}
        for _ in range(args.reps):
            for k in (0, 1):
                tune(23, k)
                best[k].append(time_launches(fe, wav, out, args.n))
        tune(23, 0)
        print(f"{win}/{hop} {method}: pk3 {min(best[0]):.1f} us (median {sorted(best[0])[len(best[0]) // 2]:.1f})  pk {min(best[1]):.1f} us "
              f"(median {sorted(best[1])[len(best[1]) // 2]:.1f})  bitwise {bool(torch.equal(outs[0], outs[1]))} "
              f"maxdiff {float((outs[0] - outs[1]).abs().max()):.3g}", flush=True)
        if args.sweep and method == "mfcc":
            max_rounds = 8 if win == 640 else 4
            for k in (0, 1):
                tune(23, k)
                row = []
                for r in range(max(1, max_rounds // 2), max_rounds + 1):
                    tune(1, 10 + r)
                    time_launches(fe, wav, out, 5)
                    row.append((r, min(time_launches(fe, wav, out, args.n) for _ in range(3))))
                tune(1, 0)
                print(f"   knob23={k} rounds sweep: " + "  ".join(f"r{r}: {t:.1f}" for r, t in row), flush=True)
            tune(23, 0)
            for cap in (256, 512, 640, 768):
                tune(13, cap)
                time_launches(fe, wav, out, 5)
                t = min(time_launches(fe, wav, out, args.n) for _ in range(3))
                print(f"   pk3 grid cap {cap}: {t:.1f}", flush=True)
            tune(13, 0)
            for st in (1, 2, 3, 4, 6, 8, 12, 16):
                tune(24, st)
                time_launches(fe, wav, out, 5)
                t = min(time_launches(fe, wav, out, args.n) for _ in range(3))
                print(f"   pk3 stagger {st}: {t:.1f}", flush=True)
            tune(24, 0)


if __name__ == "__main__":
    main()
