#!/usr/bin/env python
"""bench.py -- utterances/s of the TC-ResNet keyword-spotting hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" is one eval-mode pass of the hot path over one batch of synthetic 1 s @ 16 kHz waveforms
(waveform -> MFCC -> TCResNet8-1.0 -> softmax), BASELINE.json configs[1]: batch 4096 per GPU, 40/20 ms
front-end (49x40 MFCC, the shape BASELINE.json names).  Inputs are resident in HBM before the timed region.
N > 1 shards utterances across ranks with no collective (weak scaling).  The same JSON line also carries the
training step (configs[2]: train-mode BN + backward + momentum, with the RCCL gradient all-reduce when N > 1),
the 30/10 ms front-end variant, the roofline of the dominant kernel and the CPU baseline.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

BATCH = 4096
HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)
FP32_PEAK_TFLOPS = 157.3         # f32 vector == f32-input MFMA peak
# algorithmic work per utterance (BASELINE.md section 2 / SURVEY App. B)
WORK = {
    "4020": {"win": 640, "hop": 320, "frames": 49, "mfcc_flops": 1707503.0, "net_flops": 1585344.0},
    "3010": {"win": 480, "hop": 160, "frames": 98, "mfcc_flops": 1851906.0, "net_flops": 3045312.0},
}


def synth_batch(batch: int, device, seed: int, start: int = 0) -> torch.Tensor:
    """uniform(-0.5, 0.5) noise + per-utterance sine 440*(1 + i mod 8) Hz at 0.25, built on device."""
    g = torch.Generator(device=device).manual_seed(seed)
    noise = (torch.rand((batch, 16000), generator=g, device=device, dtype=torch.float32) - 0.5)
    t = torch.arange(16000, device=device, dtype=torch.float32) / 16000.0
    f = 440.0 * (1 + (torch.arange(batch, device=device) + start) % 8).to(torch.float32)
    return (noise + 0.25 * torch.sin(2.0 * torch.pi * f[:, None] * t[None, :])).contiguous()


# TCR_BENCH_EMU=<tests/emu/_build/libtcr_emu.so>: a CPU REHEARSAL of this file's control flow (rank launch, process group, barriers,
# collectives, the JSON line) on the host-emulator build of the kernel sources -- test infrastructure (tests/test_distributed.py), never
# a measurement: the line it prints says so ("rehearsal").  Without it the gfx950 library and a GPU are required.
EMU = os.environ.get("TCR_BENCH_EMU") or None
LEGS = ("latency", "train", "train14", "train14_3010", "forward_3010", "dscnn_forward", "dscnn_train", "train_3010", "augment")


def sync():
    if EMU is None:
        torch.cuda.synchronize()


class _HostEvent:
    """stand-in for torch.cuda.Event in the CPU rehearsal (the emulator runs launches synchronously)"""
    def record(self):
        self.t = time.perf_counter()

    def elapsed_time(self, other):
        return (other.t - self.t) * 1e3


def new_event():
    return _HostEvent() if EMU else torch.cuda.Event(enable_timing=True)


def timed(fn, steps: int, warmup: int, dist_on: bool):
    import torch.distributed as dist
    for _ in range(warmup):
        fn()
    sync()
    if dist_on:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    sync()
    if dist_on:
        dist.barrier()
    sync()
    dt = time.perf_counter() - t0
    if dist_on:
        t = torch.tensor([dt], device="cpu" if EMU else "cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t[0])
    return dt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--prewarm", type=int, default=100, help="untimed launches of the headline step BEFORE the warm-up steps: the first ~50 "
                    "launches on an idle GPU run ~15 %% slower while the clocks ramp, whatever --warmup says (reported as pre_warm_launches)")
    ap.add_argument("--batch", type=int, default=BATCH, help="utterances per GPU per step (BASELINE config: 4096)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the train / 3010 legs (profiling runs); same as --legs none")
    ap.add_argument("--legs", default="all", help="secondary legs to run next to the headline: 'all', 'none' or a comma list of " + ", ".join(LEGS))
    args = ap.parse_args()
    if args.no_extras or args.legs == "none":
        legs = set()
    elif args.legs == "all":
        legs = set(LEGS) if int(os.environ.get("WORLD_SIZE", "1")) == 1 and args.gpus == 1 and os.environ.get("TCR_BENCH_FORCE_DIST") != "1" else {"train", "train14", "train14_3010"}      # (several ranks: the two legs whose gradient all-reduce crosses xGMI -- configs[2] and configs[3], TCResNet14-1.5 at global batch 4096 N; see "secondary legs" below)
    else:
        legs = set(x for x in args.legs.split(",") if x)
        if legs - set(LEGS):
            ap.error(f"--legs: unknown {sorted(legs - set(LEGS))}; known: {', '.join(LEGS)}")

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus))          # plain `python bench.py --gpus N`: become the launcher of N ranks

    import torch.distributed as dist
    import tcresnet_amd as T
    from tcresnet_amd.parallel import DataParallel

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {args.gpus} (or unset WORLD_SIZE and let bench.py launch its ranks)")
    # (TCR_BENCH_FORCE_DIST=1 with WORLD_SIZE=1: the process group, barriers and all-reduces of the N > 1 path in a group of one rank --
    #  how a 1-GPU box sends them through RCCL; tests/test_distributed.py)
    dist_on = world > 1 or os.environ.get("TCR_BENCH_FORCE_DIST") == "1"
    if EMU:
        lib = T._lib.load_from(EMU, "emu")
        dev = torch.device("cpu")
        if dist_on:
            dist.init_process_group("gloo")
    else:
        lib = None                                            # the gfx950 library (T._lib.get() raises when it is missing)
        assert torch.cuda.is_available(), "bench.py needs a GPU (there is no CPU fallback for the product path)"
        if world > torch.cuda.device_count() and os.environ.get("TCR_BENCH_BACKEND", "nccl") == "nccl":
            raise SystemExit(f"--gpus {world} needs {world} visible GPUs for RCCL (one rank per GPU); {torch.cuda.device_count()} visible. "
                             "TCR_BENCH_BACKEND=gloo rehearses the control flow with the ranks time-sharing the visible devices.")
        local_rank %= max(torch.cuda.device_count(), 1)       # (lets a 1-GPU box rehearse the N > 1 control flow over gloo)
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
        if dist_on:
            backend = os.environ.get("TCR_BENCH_BACKEND", "nccl")    # "nccl" == RCCL over xGMI; "gloo" only for rehearsals
            if backend == "nccl":
                dist.init_process_group("nccl", device_id=dev)
            else:
                dist.init_process_group(backend)

    B = args.batch
    from tcresnet_amd.parallel import ranks_share_gpu as ranks_share_gpu_
    # what the collective of the training legs actually is: dist.get_backend() ("nccl" on ROCm is RCCL over xGMI)
    coll = {"nccl": "RCCL (backend nccl)", "gloo": "gloo (CPU rehearsal, not RCCL)"}.get(dist.get_backend(), dist.get_backend()) if dist_on else None

    def build(tag):
        w = WORK[tag]
        fe = T.Frontend(window_size_samples=w["win"], window_stride_samples=w["hop"], lib=lib, device=dev)
        net = T.TCResNet("TCResNet8", [16, 24, 32, 48], 40, fe.n_frames, 12, lib=lib, device=dev)
        net.init_xavier(0)
        return fe, net

    wav = synth_batch(B, dev, 1234 + rank, start=rank * B)
    labels = torch.zeros((B, 12), device=dev)
    labels[torch.arange(B), (torch.arange(B) + rank * B) % 12] = 1.0

    # ---------------- headline: eval forward, 49x40 front-end ----------------
    # One step = one batch through the fused MFCC kernel + the whole-network fused kernel.  Whole batches ALTERNATE between three streams
    # that have hardware queues of their own (tcresnet_amd.pipeline.InferencePipeline, mode "alternate", ways 3: batch k's two kernels back
    # to back on one stream, batch k+1's on the next, three batches in flight -- 256 us per batch against 261 with two streams, 267 with four; every step computes its own batch, outputs bitwise the one-stream sequence's:
    # tests/test_gpu_parity.py::test_inference_pipeline_equals_sequential) -- 8-9 % more batches per second than the one-stream
    # sequence, which is timed first (its two HIP-event intervals add up to its step, and its front-end interval is the dominant
    # kernel's SOLO duration: co-running, each kernel stretches to ~260 us and no longer describes itself).
    fe, net = build("4020")
    feat = torch.empty((B, 40, fe.n_frames + 8), device=dev)
    outbuf = (torch.empty((B, 12), device=dev), torch.empty((B, 12), device=dev))
    w = WORK["4020"]
    EV_EVERY = 8        # an event record between two kernels costs ~4 us of dispatch gap (281 us/step without, 292 with three per step): sampled

    def seq_step(e=None):
        if e is None:
            fe(wav, out=feat)
            net.forward_infer(feat, out=outbuf)
            return
        e[0].record()
        fe(wav, out=feat)
        e[1].record()
        net.forward_infer(feat, out=outbuf)
        e[2].record()

    # (0) everything that idles the GPU (allocations, the one-off probing of the library's internal streams) happens BEFORE the pre-warm:
    #     an idle gap of tens of ms in front of the timed region lets the clocks fall again (driver flags --steps 20 --warmup 5: 0.321 ms
    #     per step with the pipeline built between the sequence and the warm-up steps, 0.26 built here)
    pipelined = not EMU and not ranks_share_gpu_()
    if pipelined:
        from tcresnet_amd.pipeline import InferencePipeline
        pipe = InferencePipeline(fe, net, B, mode="alternate", ways=int(os.environ.get("TCR_BENCH_WAYS", "3")))     # (2 / 4: A/B arms)
        step_out = pipe.out
    # (0) the COLD figure: the headline schedule's first 20 steps after 5 warm-up steps, ahead of every pre-warm launch of
    #     this process -- what `--steps 20 --warmup 5 --prewarm 0` sees on an idle GPU whose clocks have not ramped.  Reported, never `value`.
    cold_ms = None
    if not EMU and args.prewarm > 0:
        cold_fn = (lambda: pipe.submit(wav)) if pipelined else (lambda: seq_step())
        dt_cold = timed(cold_fn, 20, 5, dist_on)
        cold_ms = round(dt_cold / 20 * 1e3, 4)
    # (1) clock pre-warm + the one-stream sequence: labelled, untimed by the contract (outside the K timed steps and the W warm-up steps).
    #     The first ~50 launches on an idle GPU run ~15 % slower whatever --warmup says; the sequence's sampled events give the solo kernels.
    for _ in range(max(0, args.prewarm)):
        seq_step()
    sync()
    nseq = 0 if EMU else max(64, args.steps)
    sev = {i: [new_event() for _ in range(3)] for i in range(nseq) if i % EV_EVERY == 2}
    c_seq = [0]

    def seq_counted():
        seq_step(sev.get(c_seq[0]))
        c_seq[0] += 1

    dt_seq = timed(seq_counted, nseq, 0, dist_on) if nseq else 0.0
    # (2) the headline: K timed steps after W warm-up steps (behind a labelled, untimed pre-warm of the multi-stream schedule itself)
    if pipelined:
        for _ in range(max(0, args.prewarm) // 2):
            pipe.submit(wav)
    nev = args.steps + args.warmup
    ev = {i: [new_event() for _ in range(3)] for i in range(args.warmup, nev) if (i - args.warmup) % EV_EVERY == 2 or (args.steps <= 2 and i == args.warmup)}       # timed steps 2, 10, 18, ...
    counter = [0]

    def fwd_step():
        i = counter[0]
        counter[0] += 1
        if pipelined:
            pipe.submit(wav, events=ev.get(i))
        else:
            seq_step(ev.get(i))

    dt = timed(fwd_step, args.steps, args.warmup, dist_on)
    value = world * B * args.steps / dt
    timed_ev = [ev[i] for i in range(args.warmup, args.warmup + args.steps) if i in ev]
    fe_in = sum(e[0].elapsed_time(e[1]) for e in timed_ev) / len(timed_ev)          # inside the timed region (co-running when pipelined)
    net_in = sum(e[1].elapsed_time(e[2]) for e in timed_ev) / len(timed_ev)
    per_step = sorted(e[0].elapsed_time(e[2]) for e in timed_ev)
    pct = lambda q: round(per_step[min(len(per_step) - 1, int(q * len(per_step)))], 4)
    if sev:
        fe_ms = sum(e[0].elapsed_time(e[1]) for e in sev.values()) / len(sev)       # solo: the one-stream sequence
        net_ms = sum(e[1].elapsed_time(e[2]) for e in sev.values()) / len(sev)
    else:
        fe_ms, net_ms = fe_in, net_in
    bitwise = None
    if pipelined:
        sync()
        ref_logits = net.forward_infer(fe(wav))[0]
        bitwise = bool(all(torch.equal(o[0], ref_logits) for o in step_out))

    # dominant kernel = the fused front-end (one launch per step): waveform read once, [40][49] tile written once
    fe_bytes = B * (16000 * 4 + 40 * w["frames"] * 4)
    fe_flops = B * w["mfcc_flops"]
    fe_gbs = fe_bytes / (fe_ms * 1e-3) / 1e9
    fe_tf = fe_flops / (fe_ms * 1e-3) / 1e12
    hbm_frac, fp_frac = fe_gbs / HBM_PEAK_GBS, fe_tf / FP32_PEAK_TFLOPS
    if fp_frac >= hbm_frac:
        # ("valu": the kernel's flops issue on the packed-FP32 vector pipe, not on the matrix cores -- the f32 peak of both pipes is the
        #  same 157.3 TFLOP/s, so `peak` is what the contract's "mfma" bound would carry)
        roof = {"bound": "valu", "achieved": round(fe_tf, 3), "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(fp_frac, 4)}
    else:
        roof = {"bound": "hbm", "achieved": round(fe_gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(hbm_frac, 4)}
    traffic, traffic_src = pmc_traffic("frontend_pk3_kernel<512,")
    prof_avg_us, prof_src = profile_avg_us("frontend_pk3_kernel<512,")
    roof.update({"traffic": traffic, "traffic_source": traffic_src, "profile_avg_us": prof_avg_us, "profile_source": prof_src, "kernel": "frontend_pk3_kernel<512, 10, false>", "kernel_ms": round(fe_ms, 4),
                 "kernel_ms_in_timed_region": round(fe_in, 4),
                 "note": "f32 VALU pipe; its 157.3 TFLOP/s peak equals the exact-f32 MFMA peak. kernel_ms: HIP-event-bracketed launches on the launch "
                         "stream in the one-stream sequence of this run (the kernel alone on the chip; every 8th step: an event record costs ~4 us of "
                         "dispatch gap); kernel_ms_in_timed_region: the same brackets inside the timed region, where the kernel shares the chip with the "
                         "other stream's network kernel -- see pair_roofline for the co-running pair",
                 "hbm_gbs": round(fe_gbs, 1), "hbm_frac": round(hbm_frac, 4), "fp32_tflops": round(fe_tf, 3), "fp32_frac": round(fp_frac, 4),
                 "algorithmic_bytes_per_launch": fe_bytes, "algorithmic_flops_per_launch": fe_flops})
    if prof_avg_us and abs(prof_avg_us - fe_ms * 1e3) > 0.05 * fe_ms * 1e3:
        roof["profile_stale_warning"] = (f"committed profile says {prof_avg_us} us per launch, this run measured {round(fe_ms * 1e3, 1)} us (> 5 % apart): "
                                         f"{prof_src} (and the PMC traffic beside it) describes another build or another clock state")
    whole_tf = value / world * (w["mfcc_flops"] + w["net_flops"]) / 1e12
    out = {
        "metric": "utterances/sec (1 s@16 kHz) TCResNet8-1.0 forward", "value": round(value, 1), "unit": "utterances/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"TCResNet8-1.0 eval forward, waveform->softmax, batch {B}/GPU, 49x40 MFCC (40/20 ms, FFT 1024), 12 classes",
                   "global_batch": world * B, "parallelism": f"dp{world} (utterance shards, no collective)", "collective_backend": dist.get_backend() if dist_on else None,
                   "schedule": "whole batches alternating between three streams, three in flight (InferencePipeline 'alternate', ways 3)" if pipelined else "one stream"},
        "collective_backend": coll, "collectives_per_step": {"forward": 0},       # (eval forward: replicas only; the training legs add theirs below)
        "roofline": roof,
        # the co-running pair: algorithmic flops of BOTH kernels of a step / the step time of the timed region
        "pair_roofline": {"bound": "mfma", "achieved": round(whole_tf, 3), "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(whole_tf / FP32_PEAK_TFLOPS, 4),
                          "kernels": ["frontend_pk3_kernel<512, 10, false>", "net_fused_tc8_kernel<8, 49, 0, 3>"],
                          "algorithmic_flops_per_step": B * (w["mfcc_flops"] + w["net_flops"])},
        "phases_ms": {"frontend": round(fe_ms, 4), "net": round(net_ms, 4), "frontend_in_timed_region": round(fe_in, 4), "net_in_timed_region": round(net_in, 4)},
        "sequential": {"ms_per_step": round(dt_seq / nseq * 1e3, 4) if nseq else None, "steps": nseq,
                       "what": "the same batches on ONE stream (front-end, network back to back), timed in this run ahead of the warm-up steps"},
        "bitwise_equal_to_sequential": bitwise,
        ("batch_latency_ms_p10_p50_p90" if pipelined else "step_ms_p10_p50_p90"): [pct(0.1), pct(0.5), pct(0.9)],     # (pipelined: e0 -> e2 of a batch on its own stream, three batches in flight)
        "event_timed_steps": len(timed_ev),
        "pre_warm_launches": max(0, args.prewarm) + (max(0, args.prewarm) // 2 if pipelined else 0),
        "cold_ms_per_step": cold_ms,        # (steps 6..25 of this process's headline schedule, ahead of the pre-warm; None with --prewarm 0, where ms_per_step IS the cold figure)
        "whole_path_fp32_frac": round(whole_tf / FP32_PEAK_TFLOPS, 4),
        "whole_path_hbm_frac": round(value / world * 64048 / 1e9 / HBM_PEAK_GBS, 4),
    }

    if EMU:
        out["rehearsal"] = "CPU emulator build of the kernel sources (TCR_BENCH_EMU): control flow only, NOT a measurement"
        out["data"] = "synthetic (CPU rehearsal)"

    # ---------------- secondary legs ----------------
    # The headline above is complete at this point.  The legs below never ran on more than one GPU before the driver's scaling runs, so
    # with several ranks (a) only the training leg runs by default -- the one whose gradient all-reduce crosses xGMI --, (b) an exception
    # in a leg is recorded as {"error": ...} under "secondary_legs" instead of losing the line, and (c) a watchdog prints the line as it
    # stands and ends the rank if the legs have not finished in TCR_BENCH_LEG_TIMEOUT seconds (a hung collective cannot be caught).
    import threading
    legs_done = threading.Event()

    def emit():
        if rank == 0:
            print(json.dumps(out), flush=True)

    def watchdog():
        limit = float(os.environ.get("TCR_BENCH_LEG_TIMEOUT", "240"))
        if not legs_done.wait(limit):
            out["secondary_legs"] = {"error": f"not finished after {limit:.0f} s: the line is the headline plus the legs completed so far"}
            emit()
            os._exit(0)

    if dist_on and legs:
        threading.Thread(target=watchdog, daemon=True).start()

    def secondary():
        if os.environ.get("TCR_BENCH_FAIL_LEG") == "raise":        # (tests: the two failure paths above)
            raise RuntimeError("injected leg failure")
        if os.environ.get("TCR_BENCH_FAIL_LEG") == "hang":
            time.sleep(3600)
        from tcresnet_amd.pipeline import FeaturePrefetcher
        from tcresnet_amd.parallel import ranks_share_gpu
        overlap = EMU is None and not ranks_share_gpu()
        step_no = [0]
        # every secondary leg: >= 20 timed steps after >= 10 warm-up steps whatever the command line says (the driver's
        # --steps 20 --warmup 5 used to leave the DS-CNN leg 6 steps after 2: a bimodal 4.3 / 8.8 ms); the CPU rehearsal keeps the flags
        floor = (lambda lo, v: v) if EMU else (lambda lo, v: max(lo, v))
        tsteps, twarm = floor(20, args.steps // 2), floor(10, args.warmup // 2)

        def train_leg(fe_, net_, steps, warm):
            """One training leg: every step computes the MFCC of its own batch; like the reference's tf.data prefetch, the front-end of step
            k+1 is issued on a second stream while step k's forward/backward/update occupy the main stream (FeaturePrefetcher)."""
            dp = DataParallel(net_)
            pf = FeaturePrefetcher(fe_, B, overlap=overlap)
            pf.submit(wav)
            early = FeaturePrefetcher.submit_point(net_) == "before_forward"      # (measured per net family: pipeline.py)

            def step():
                step_no[0] += 1
                f = pf.get()
                if early: pf.submit(wav, input_ready=True)      # (the synthetic batch was written and synchronised before the legs began)
                dp.forward_train(f, labels, keep_prob=0.5, seed=step_no[0])
                if not early: pf.submit(wav, input_ready=True)
                dp.backward()
                net_.sgd_momentum_step(0.1, 0.9, 0.001)

            c0 = dp.collectives
            dt_ = timed(step, steps, warm, dist_on)
            return {"value": round(world * B * steps / dt_, 1), "unit": "utterances/s", "ms_per_step": round(dt_ / steps * 1e3, 4), "steps": steps,
                    "collectives_per_step": round((dp.collectives - c0) / (steps + warm), 2)}

        if "latency" in legs:
            # ---------------- batch-1 latency (configs[0]'s regime; the CPU baseline's batch-1 number sits in cpu_baseline.forward_by_batch) ----
            w1 = wav[:1].contiguous()
            o1 = (torch.empty((1, 12), device=dev), torch.empty((1, 12), device=dev))
            for _ in range(50):
                net.forward_waveform(fe, w1, out=o1)
            sync()
            t0 = time.perf_counter()
            for _ in range(200):
                net.forward_waveform(fe, w1, out=o1)
                sync()
            lat_py = (time.perf_counter() - t0) / 200 * 1e6
            # the same C-ABI call PREPARED once (TCResNet.waveform_call: pointers bound, BN folded, no per-call Python bookkeeping) -- what a
            # C / C++ host of the boundary pays per utterance; forward_waveform's wrapper costs ~9 us of Python on top (rounds 1-5 reported that)
            call1 = net.waveform_call(fe, w1, o1)
            for _ in range(50):
                call1()
                sync()
            t0 = time.perf_counter()
            for _ in range(400):
                call1()
                sync()
            lat = (time.perf_counter() - t0) / 400 * 1e6
            out["latency_batch_1"] = {"value": round(lat, 1), "unit": "us", "higher_is_better": False,
                                      "through_the_python_wrapper_us": round(lat_py, 1),
                                      "workload": "one utterance, waveform -> softmax: ONE prepared host call of tcr_forward_waveform + device synchronisation per "
                                                  "utterance (through_the_python_wrapper_us: the same via TCResNet.forward_waveform, as rounds 1-5 reported): "
                                                  "front-end ~9 us + small-batch network kernel ~21 us (weights DMA-copied into LDS a phase ahead; "
                                                  "bitwise the throughput kernel) back to back, ~12 us of host launch + synchronisation latency"}
            # the same prepared call at batch 64 (the serving regime between one utterance and the throughput batch)
            n64 = min(64, B)
            w64 = wav[:n64].contiguous()
            o64 = (torch.empty((n64, 12), device=dev), torch.empty((n64, 12), device=dev))
            call64 = net.waveform_call(fe, w64, o64)
            for _ in range(50):
                call64()
                sync()
            t0 = time.perf_counter()
            for _ in range(400):
                call64()
                sync()
            lat64 = (time.perf_counter() - t0) / 400 * 1e6
            out["latency_batch_64"] = {"value": round(lat64, 1), "unit": "us", "higher_is_better": False, "utterances_per_s": round(n64 * 1e6 / lat64, 1),
                                       "workload": f"{n64} utterances, waveform -> softmax: one prepared tcr_forward_waveform call + device synchronisation per batch"}
        if "train" in legs:
            # ---------------- training step (configs[2]) ----------------
            out["train"] = train_leg(fe, net, tsteps, twarm)
            out["train"]["workload"] = ("TCResNet8-1.0 train step: MFCC (prefetched on a second stream) + train-mode BN fwd + bwd + momentum (wd 1e-3, keep_prob 0.5), "
                                        f"batch {B}/GPU" + (f", {coll} all-reduce of the flat gradient arena" if dist_on else ""))
            out["collectives_per_step"]["train"] = out["train"]["collectives_per_step"]
        if "train14" in legs:
            # ---------------- TCResNet14-1.5 training (configs[3]: global batch 32768 = 8 x 4096 over RCCL) ----------------
            net14 = T.TCResNet("TCResNet14", [24, 36, 36, 48, 48, 72, 72], 40, fe.n_frames, 12, lib=lib, device=dev)
            net14.init_xavier(0)
            out["train_tcresnet14_1.5"] = train_leg(fe, net14, floor(20, max(1, args.steps // 4)), floor(10, args.warmup // 2))
            out["train_tcresnet14_1.5"]["workload"] = (f"TCResNet14-1.5 train step, batch {B}/GPU (global {world * B}), 303 144 params"
                                                       + (f", {coll} all-reduce of the 1.21 MB gradient arena" if dist_on else ""))
            out["collectives_per_step"]["train_tcresnet14_1.5"] = out["train_tcresnet14_1.5"]["collectives_per_step"]
            del net14
        if legs & {"forward_3010", "train_3010", "train14_3010"}:
            fe2, net2 = build("3010")
        if "train14_3010" in legs:
            # ---------------- configs[3] at the reference's own front-end setting: the only TCResNet14-1.5 script is 30/10 ms -> 98 frames
            # (scripts/commands/TCResNet14Model-1.5_mfcc_40_3010_0.001_mom_l1.sh:3) ----------------
            net14b = T.TCResNet("TCResNet14", [24, 36, 36, 48, 48, 72, 72], 40, fe2.n_frames, 12, lib=lib, device=dev)
            net14b.init_xavier(0)
            out["train_tcresnet14_1.5_3010"] = train_leg(fe2, net14b, floor(20, max(1, args.steps // 4)), floor(10, args.warmup // 2))
            out["train_tcresnet14_1.5_3010"]["workload"] = (f"TCResNet14-1.5 train step, 98x40 MFCC (30/10 ms: the reference's script for this model), batch {B}/GPU (global {world * B})"
                                                            + (f", {coll} all-reduce of the 1.21 MB gradient arena" if dist_on else ""))
            out["collectives_per_step"]["train_tcresnet14_1.5_3010"] = out["train_tcresnet14_1.5_3010"]["collectives_per_step"]
            del net14b
        if "forward_3010" in legs:
            # ---------------- 30/10 ms front-end (98x40, the reference's training scripts) ----------------
            feat2 = torch.empty((B, 40, fe2.n_frames + 8), device=dev)

            def fwd2():
                fe2(wav, out=feat2)
                net2.forward_infer(feat2)

            s2 = floor(50, args.steps)
            dt2_seq = timed(fwd2, s2, floor(20, args.warmup), dist_on)
            dt2, sched2 = dt2_seq, "one stream"
            if pipelined:           # the headline's schedule (whole batches alternating between three streams)
                from tcresnet_amd.pipeline import InferencePipeline
                pipe2 = InferencePipeline(fe2, net2, B, mode="alternate", ways=int(os.environ.get("TCR_BENCH_WAYS", "3")))
                dt2 = timed(lambda: pipe2.submit(wav), s2, floor(20, args.warmup), dist_on)
                sched2 = "whole batches alternating between three streams (InferencePipeline 'alternate')"
                del pipe2
            out["forward_3010"] = {"value": round(world * B * s2 / dt2, 1), "unit": "utterances/s", "ms_per_step": round(dt2 / s2 * 1e3, 4), "steps": s2,
                                   "schedule": sched2, "sequential": {"ms_per_step": round(dt2_seq / s2 * 1e3, 4), "steps": s2},
                                   "workload": "same, 98x40 MFCC (30/10 ms, FFT 512)"}

        if legs & {"dscnn_forward", "dscnn_train"}:
            fe3 = T.Frontend(window_size_samples=640, window_stride_samples=320, num_mfccs=10, lib=lib, device=dev)
            ds = T.DSCNN("L", fe3.n_frames, 10, 12, lib=lib, device=dev)
            ds.init_xavier(0)
            feat3 = torch.empty((B, 10, fe3.n_frames + 8), device=dev)
            ds_flops = 2.0 * 28327812.0          # 28.33 M MAC / utterance
        if "dscnn_forward" in legs:
            # ---------------- DS-CNN-L forward (configs[4]): 49x10 MFCC, batch 4096 ----------------
            def fwd3():
                fe3(wav, out=feat3)
                ds.forward_infer(feat3)

            dsteps = floor(30, args.steps // 3)
            dt3_seq = timed(fwd3, dsteps, floor(10, args.warmup // 3), dist_on)
            # the headline's schedule for this network too: whole batches alternating between two streams (one set of activation buffers per
            # stream, the same parameters), so that one batch's kernels fill the tail rounds and the prologue / epilogue bubbles of the other's
            # (the pointwise conv is 4160 workgroups on 1024 slots: a fifth, 6 %-full round per launch); every batch still runs its own
            # front-end and network back to back on its stream
            ways3 = 1 if EMU else 2
            ds_alt = [ds] + [T.DSCNN("L", fe3.n_frames, 10, 12, lib=lib, device=dev) for _ in range(ways3 - 1)]
            for d2 in ds_alt[1:]:
                d2.load_state_dict(ds.state_dict())
            feat_alt = [feat3] + [torch.empty_like(feat3) for _ in range(ways3 - 1)]
            str_alt = [torch.cuda.Stream() for _ in range(ways3)] if not EMU else [None]
            turn = [0]

            def fwd3_alt():
                i = turn[0] % ways3
                turn[0] += 1
                if EMU:
                    fe3(wav, out=feat_alt[i]); ds_alt[i].forward_infer(feat_alt[i])
                    return
                with torch.cuda.stream(str_alt[i]):
                    fe3(wav, out=feat_alt[i])
                    ds_alt[i].forward_infer(feat_alt[i])

            if not EMU:
                for st in str_alt:
                    st.wait_stream(torch.cuda.current_stream())
            dt3 = timed(fwd3_alt, dsteps, floor(10, args.warmup // 3), dist_on)
            if not EMU:
                for st in str_alt:
                    torch.cuda.current_stream().wait_stream(st)
            same3 = bool(torch.equal(ds_alt[0].forward_infer(feat_alt[0])[0], ds_alt[-1].forward_infer(feat_alt[-1])[0]))
            out["dscnn_l_forward"] = {"value": round(world * B * dsteps / dt3, 1), "unit": "utterances/s", "ms_per_step": round(dt3 / dsteps * 1e3, 4),
                                      "steps": dsteps, "net_tflops": round(world * B * dsteps / dt3 * ds_flops / 1e12 / world, 2),
                                      "schedule": f"whole batches alternating between {ways3} streams",
                                      "sequential": {"ms_per_step": round(dt3_seq / dsteps * 1e3, 4), "steps": dsteps,
                                                     "what": "the same batches on ONE stream (rounds 1-4 reported this)"},
                                      "streams_agree_bitwise": same3,
                                      "workload": f"DSCNNLModel eval forward, waveform->softmax, 49x10 MFCC, batch {B}/GPU"}

        if "dscnn_train" in legs:
            # ---------------- DS-CNN-L training step (configs[4], training half): Adam lr 5e-4 ----------------
            dpd = DataParallel(ds)
            ds_step = [0]

            def train_ds():
                ds_step[0] += 1
                fe3(wav, out=feat3)             # in line: prefetched next to the step (FeaturePrefetcher) it costs more than its own 0.17 ms --
                dpd.forward_train(feat3, labels)    # 16.89 against 16.15 ms per step: its 2 x 77 KB of LDS per CU displace the step's LDS-tiled kernels
                dpd.backward()
                ds.adam_step(5e-4, ds_step[0])

            tds = floor(20, args.steps // 6)
            c0 = dpd.collectives
            dtd = timed(train_ds, tds, floor(10, args.warmup // 2), dist_on)
            out["dscnn_l_train"] = {"value": round(world * B * tds / dtd, 1), "unit": "utterances/s", "ms_per_step": round(dtd / tds * 1e3, 4),
                                    "steps": tds, "net_tflops": round(B * tds / dtd * 3.0 * ds_flops / 1e12, 2),
                                    "workload": f"DSCNNLModel train step: MFCC + train-mode BN fwd + bwd + Adam, batch {B}/GPU"
                                                + (f", {coll} all-reduce of the gradient arena" if dist_on else "")}
        if "train_3010" in legs:
            # ---------------- TCResNet8 training with the 30/10 ms front-end (the reference's training scripts) ----------------
            out["train_3010"] = train_leg(fe2, net2, tsteps, twarm)
            out["train_3010"]["workload"] = f"TCResNet8-1.0 train step, 98x40 MFCC (30/10 ms), batch {B}/GPU"

        if "augment" in legs:
            # ---------------- input stage: PCM16 -> shift -> background mix -> clip, batch 4096 ----------------
            alib = net.lib
            gen = torch.Generator(device=dev).manual_seed(7 + rank)
            pcm = torch.randint(-32768, 32768, (B * 16000,), generator=gen, device=dev, dtype=torch.int32).to(torch.int16)
            bgp = torch.randint(-32768, 32768, (6 * 60 * 16000,), generator=gen, device=dev, dtype=torch.int32).to(torch.int16)
            clip_off = (torch.arange(B, device=dev, dtype=torch.int64) * 16000).contiguous()
            clip_len = torch.full((B,), 16000, device=dev, dtype=torch.int32)
            shift = torch.randint(-1600, 1600, (B,), generator=gen, device=dev, dtype=torch.int32)
            bg_off = torch.randint(0, 6 * 60 * 16000 - 16000, (B,), generator=gen, device=dev, dtype=torch.int64)
            mixed = torch.rand((B,), generator=gen, device=dev) < 0.8
            bg_vol = (torch.rand((B,), generator=gen, device=dev) * 0.1 * mixed).contiguous()
            aug_out = torch.empty((B, 16000), device=dev)
            stream = None if EMU else torch.cuda.current_stream(dev).cuda_stream

            def aug():
                alib.check(alib.tcr_augment_fwd(pcm.data_ptr(), clip_off.data_ptr(), clip_len.data_ptr(), shift.data_ptr(), bgp.data_ptr(),
                                                bg_off.data_ptr(), bg_vol.data_ptr(), B, 16000, aug_out.data_ptr(), stream), "tcr_augment_fwd")

            sa = floor(50, args.steps)
            dta = timed(aug, sa, floor(20, args.warmup), dist_on)
            aug_bytes = B * 16000 * (2 + 4) + int(mixed.sum().item()) * 16000 * 2
            out["augment"] = {"value": round(world * B * sa / dta, 1), "unit": "utterances/s", "ms_per_step": round(dta / sa * 1e3, 4), "steps": sa,
                              "hbm_gbs": round(aug_bytes / (dta / sa) / 1e9, 1), "hbm_frac": round(aug_bytes / (dta / sa) / 1e9 / HBM_PEAK_GBS, 4),
                              "algorithmic_bytes_per_launch": aug_bytes,
                              "workload": f"tcr_augment_fwd: int16 PCM -> float, +-1600-sample shift, background mix (80 % of utterances), clip; batch {B}/GPU"}

    try:
        secondary()
    except Exception as e:                          # (one rank: let it surface -- the GPU tests and the driver's N = 1 run want the traceback)
        if not dist_on:
            raise
        out["secondary_legs"] = {"error": f"{type(e).__name__}: {e}"[:400]}
    legs_done.set()
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not EMU:
        out["cpu_baseline"] = cpu_baseline()
    emit()
    if dist_on:
        dist.destroy_process_group()


def self_launch(n: int) -> int:
    """`python bench.py --gpus N` without a launcher's environment: re-run this command under
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port <free>` -- one rank per GPU,
    RCCL over xGMI (backend "nccl") --, pass the ranks' output through (rank 0 prints the ONE JSON line) and return their exit code.
    The rendezvous is on 127.0.0.1 (the container's hostname may not resolve); HSA_ENABLE_IPC_MODE_LEGACY=0 is kept/added because
    the host driver only supports dmabuf IPC, which RCCL's intra-node transport needs."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")             # (torch.distributed.run would set 1 and print a warning block on stderr)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def pmc_traffic(kernel: str):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes of this same command (scripts/gpu_prof.sh ->
    profiles/*_pmc.csv): FETCH_SIZE x 2 (gfx950 counts 128-byte requests at 64 B, MI355X_MICROARCH.md "HBM") + WRITE_SIZE,
    both in KiB.  None when no profile is committed."""
    import csv
    import glob
    paths = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "*_pmc.csv")), reverse=True)
    for path in sorted(paths, key=lambda q: "_final_" not in os.path.basename(q)):      # the round's final profile first (stable sort)
        vals = {}
        for r in csv.DictReader(open(path)):
            if r["kernel"].startswith(kernel) and r["counter"] in ("FETCH_SIZE", "WRITE_SIZE"):
                vals[r["counter"]] = float(r["mean_per_launch"])
        if len(vals) == 2:
            return int((2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024), os.path.join("profiles", os.path.basename(path))
    return None, None


def profile_avg_us(kernel: str):
    """Average launch duration of `kernel` in the committed forward-only rocprofv3 --kernel-trace --stats summary of this command
    (`bench.py --no-extras --no-cpu-baseline` under scripts/gpu_prof.sh -> profiles/*_fwd_kernel_stats.csv): the file the
    roofline's `kernel_ms` can be recomputed from."""
    import csv
    import glob
    paths = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "*_fwd_kernel_stats.csv")), reverse=True)
    for path in sorted(paths, key=lambda q: "_final_" not in os.path.basename(q)):      # the round's final profile first (stable sort)
        for r in csv.DictReader(open(path)):
            if r["kernel"].startswith(kernel):
                return round(float(r["avg_ns"]) / 1e3, 2), os.path.join("profiles", os.path.basename(path))
    return None, None


def cpu_baseline():
    """The oracle's PyTorch-CPU float32 restatement of the same path, timed on the host cores (bounded sample, ~25 s):
    eval forward at batch 1 (BASELINE.json configs[0]), 256 and 4096, and one training step (fwd + autograd bwd + momentum) at 256."""
    from oracle import numpy_ref as R
    from oracle import torch_ref as TR
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    arch = R.make_tcresnet("TCResNet8", 1.0)
    p, s = R.init_params(arch, 0)
    probe = torch.from_numpy(R.synth_waveforms(64))
    best, threads = 0.0, 1
    for th in sorted({min(avail, c) for c in (8, 16, 32, 64, 128)}):      # oneDNN/OpenMP oversubscribe badly on tiny convs
        cb = TR.CpuBaseline(arch, R.FRONTEND_4020, p, s, th)
        cb.infer(probe)
        t0, n = time.perf_counter(), 0
        while time.perf_counter() - t0 < 1.0:
            cb.infer(probe)
            n += 1
        rate = n * 64 / (time.perf_counter() - t0)
        if rate > best:
            best, threads = rate, th
    cb = TR.CpuBaseline(arch, R.FRONTEND_4020, p, s, threads)

    def rate_of(fn, b, budget):
        fn()
        n, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < budget:
            fn()
            n += 1
        dt = time.perf_counter() - t0
        return round(n * b / dt, 1), n, dt

    legs = {}
    base256 = torch.from_numpy(R.synth_waveforms(256))
    for b, budget in ((1, 3.0), (256, 6.0), (4096, 6.0)):
        wav = base256[:b] if b <= 256 else base256.repeat(b // 256, 1)
        legs[b] = rate_of(lambda: cb.infer(wav), b, budget)
    lab = torch.from_numpy(R.synth_labels(256).astype("float32"))
    tr = rate_of(lambda: cb.train_step(base256, lab, 0.1, 0.9, 0.001), 256, 5.0)
    v, n, dt = legs[256]
    return {"value": v, "unit": "utterances/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{n} x batch 256 eval forwards (waveform->softmax, 49x40 MFCC) in {dt:.1f} s; PyTorch-CPU f32 restatement "
                      f"of the TF1 graph (oracle/torch_ref.py); {torch.get_num_threads()} threads (best of a 1 s sweep), "
                      f"{avail} usable of os.cpu_count()={os.cpu_count()}",
            "forward_by_batch": {str(b): {"value": legs[b][0], "batches": legs[b][1], "seconds": round(legs[b][2], 1)} for b in legs},
            "train_step_batch_256": {"value": tr[0], "steps": tr[1], "seconds": round(tr[2], 1),
                                     "what": "MFCC + train-mode BN forward + autograd backward + momentum update, keep_prob 1"}}


if __name__ == "__main__":
    main()
