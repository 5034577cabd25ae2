"""Two-stream inference pipeline: the front-end of batch k+1 overlaps the network of batch k.

The fused MFCC kernel is VALU/latency-bound and the network kernels run on the matrix cores, so the two
co-reside on the CUs; features and outputs are double-buffered and ordered with HIP events
(front-end(k+2) waits for network(k) before reusing its feature buffer)."""
from __future__ import annotations

from typing import List, Tuple

import torch

from ._lib import padded_len


_STREAMS = {}
_ROLE_INDEX = {"frontend": 2, "network": 3, "aux0": 0, "aux1": 1}      # (0 / 1: the training step's filter-gradient streams, idle during inference)


def shared_stream(dev, role: str) -> "torch.cuda.Stream":
    """One stream per (device, role) for the whole process: the library's internal stream of that role (`tcr_internal_stream`).
    HIP multiplexes streams onto a few hardware queues and streams that share a queue serialise -- the two-stream pipeline runs at
    272 us per batch on streams with queues of their own, 313 on two that share one (scripts/stream_concurrency_probe.py) -- so the
    library probes candidates once and keeps a set that runs concurrently with the caller's current stream and with each other."""
    from . import _lib
    dev = torch.device(dev)
    index = dev.index if dev.index is not None else torch.cuda.current_device()
    key = (dev.type, index, role)
    if key not in _STREAMS:
        with torch.cuda.device(index):
            ptr = _lib.get().tcr_internal_stream(_ROLE_INDEX[role], torch.cuda.current_stream(dev).cuda_stream)
        if not ptr:
            raise _lib.TcrError(f"tcr_internal_stream({role}): {_lib.get().tcr_last_error().decode()}")
        _STREAMS[key] = torch.cuda.ExternalStream(ptr, device=dev)
    return _STREAMS[key]


class InferencePipeline:
    """mode "handoff": the front-end stream hands every batch's features to the network stream (events between the kernels; `depth`
    feature / output buffers).  mode "alternate": whole batches alternate between the two streams -- a batch's front-end and network run
    back to back on ONE stream, with no event between them, while the other stream(s) work on the next batch(es) (`ways` = 2 .. 4 streams =
    batches in flight); each stream owns one feature buffer, one output pair and one network workspace (stream order alone protects them).  Measured at batch 4096, TCResNet8:
    sequential 284 us per batch, handoff three deep 272, alternate: see DESIGN.md section 7."""

    def __init__(self, frontend, net, batch: int, depth: int = 2, mode: str = "handoff", ways: int = 2, fe_rounds="auto"):
        if mode not in ("handoff", "alternate"):
            raise ValueError(f"InferencePipeline: unknown mode {mode!r}")
        if not 2 <= int(ways) <= 4:
            raise ValueError(f"InferencePipeline: ways = {ways} (2 .. 4 streams)")
        self.mode = mode
        if mode == "alternate":
            depth = int(ways)                           # batches in flight = streams
        self.fe, self.net, self.batch, self.depth = frontend, net, int(batch), int(depth)
        dev = frontend.device
        self.s_fe, self.s_net = shared_stream(dev, "frontend"), shared_stream(dev, "network")
        self._streams = [self.s_fe, self.s_net] + [shared_stream(dev, r) for r in ("aux0", "aux1")[:max(0, depth - 2)]] if mode == "alternate" else None
        self.feat = [torch.empty((batch, frontend.n_coef, padded_len(frontend.n_frames)), device=dev) for _ in range(depth)]
        self.out = [(torch.empty((batch, net.num_classes), device=dev), torch.empty((batch, net.num_classes), device=dev))
                    for _ in range(depth)]
        if mode == "alternate":
            self._ws = [net.new_workspace(batch, False) for _ in range(depth)]  # one per stream
        else:
            self._ws = None
            self.net.workspace(batch, False)             # allocate before the streams start
        # Frames per front-end workgroup while the kernels of other batches run next to it: the launcher's round count (tcr_tune knob
        # TCR_TUNE_FRONTEND = 10 + rounds) is fitted to the front-end running alone -- 7 rounds of 8 frames at batch 4096 --; with two or
        # three batches in flight the longest chunks (8 rounds: fewer, longer-lived workgroups) measured 254.7 against 258.2 us per batch
        # (scripts/ab_pipe_knobs.py).  "auto": 8 rounds in mode "alternate" for the 1024-point FFT; None: the launcher's choice.
        if fe_rounds == "auto":
            fe_rounds = 8 if (mode == "alternate" and frontend.cfg.nfft == 1024) else None
        self.fe_rounds = fe_rounds
        self._fe_done: List[torch.cuda.Event] = [torch.cuda.Event() for _ in range(depth)]
        self._net_done: List[torch.cuda.Event] = [torch.cuda.Event() for _ in range(depth)]
        self._k = 0
        self.fe_events = None                            # optional (start, end) timing events per step

    def submit(self, wav: torch.Tensor, events=None) -> Tuple[torch.Tensor, torch.Tensor]:
        """Enqueue one batch; returns the (logits, probs) buffers it will land in (valid after `sync()`, after `done_event(...)`, or
        after `depth` further submits have been synchronised by the caller).  events (mode "alternate"): three timing events recorded
        on the batch's stream around its two kernels (e0 | front-end | e1 | network | e2)."""
        i = self._k % self.depth
        cur = torch.cuda.current_stream(self.fe.device)
        if self.mode == "alternate":
            st = self._streams[i]
            with torch.cuda.stream(st):
                st.wait_stream(cur)                             # the caller produced `wav` on its current stream
                if events is not None:
                    events[0].record(st)
                self._frontend(wav, self.feat[i])
                if events is not None:
                    events[1].record(st)
                self.net.forward_infer(self.feat[i], out=self.out[i], workspace=self._ws[i])
                if events is not None:
                    events[2].record(st)
                self._net_done[i].record(st)
            self._k += 1
            return self.out[i]
        with torch.cuda.stream(self.s_fe):
            self.s_fe.wait_stream(cur)                          # the caller produced `wav` on its current stream
            if self._k >= self.depth:
                self.s_fe.wait_event(self._net_done[i])         # feature buffer i is free again
            if self.fe_events is not None:
                self.fe_events[self._k][0].record(self.s_fe)
            self.fe(wav, out=self.feat[i])
            if self.fe_events is not None:
                self.fe_events[self._k][1].record(self.s_fe)
            self._fe_done[i].record(self.s_fe)
        with torch.cuda.stream(self.s_net):
            self.s_net.wait_event(self._fe_done[i])
            self.net.forward_infer(self.feat[i], out=self.out[i])
            self._net_done[i].record(self.s_net)
        self._k += 1
        return self.out[i]

    def _frontend(self, wav, out):
        # (a per-call launch argument, tcr_frontend_fwd_rounds: nothing process-wide is touched, other threads' / front-ends' launches and
        #  a user's TCR_TUNE_FRONTEND setting are unaffected)
        return self.fe(wav, out=out, rounds=int(self.fe_rounds or 0))

    def done_event(self, slot: int) -> "torch.cuda.Event":
        """The event recorded behind the network of the batch last submitted into output slot `slot` (= submit index % depth)."""
        return self._net_done[slot]

    def sync(self):
        cur = torch.cuda.current_stream(self.fe.device)
        for st in (self._streams or (self.s_fe, self.s_net)):
            cur.wait_stream(st)


class FeaturePrefetcher:
    """Training-side input prefetch: the front-end of batch k+1 runs on its own stream while the optimisation step of
    batch k occupies the caller's stream (the device-side counterpart of the reference's tf.data `prefetch`,
    datasets/data_wrapper_base.py:70-76).  The kernels of a training step are individually too short to fill the chip, so
    the MFCC kernel slots in between them.

        pf = FeaturePrefetcher(frontend, batch)
        pf.submit(first_wavs)
        for step in ...:
            feat = pf.get()                 # features of this step (ready on the caller's stream)
            pf.submit(next_wavs)            # FeaturePrefetcher.submit_point(net) == "before_forward": the next step's features overlap this whole step
            net.forward_train(feat, ...)
            net.backward(); net.sgd_momentum_step(...)
    """

    @staticmethod
    def submit_point(net) -> str:
        """Where in the step the next batch's front-end is best issued: in front of the forward, for every net family.  Measured at
        batch 4096 (scripts/ab_prefetch_point.py, round 3, with the front-end on a stream that has a hardware queue of its own --
        `shared_stream`): TCResNet8-1.0 1108 us against 1193 behind the forward, 1216 behind the backward, 1181 with no overlap, 995 for
        the step alone; 98 frames 1705 / 1745 / 1758 / 1730; TCResNet14-1.5 3034 / 3079 / 3104 / 3061; 98 frames 4783 / 4841 / 4853 /
        4815.  (An earlier measurement had the wide nets "after_forward": 3083 against 3126 -- taken on a stream that, as it turned out,
        shared a hardware queue with one of the backward's; the argument `net` stays for callers that ask per net.)"""
        return "before_forward"

    def __init__(self, frontend, batch: int, overlap: bool = True):
        """overlap=False degrades to the caller's stream (same results, no second stream)."""
        self.fe = frontend
        dev = frontend.device
        self.stream = shared_stream(dev, "frontend") if overlap else None
        self.feat = [torch.empty((batch, frontend.n_coef, padded_len(frontend.n_frames)), device=dev) for _ in range(2)]
        self._ready = [torch.cuda.Event(), torch.cuda.Event()]
        self._free = [None, None]           # event after which buffer i may be overwritten
        self._k = 0                         # submits so far
        self._pending = None

    def submit(self, wav: torch.Tensor, input_ready: bool = False):
        """input_ready: the caller guarantees `wav` is complete (written and synchronised earlier): the front-end stream then does not
        order itself behind the caller's current stream -- that ordering is an event record on the TRAINING stream, a few microseconds
        of dispatch gap in every step."""
        i = self._k % 2
        if self.stream is None:
            self.fe(wav, out=self.feat[i])
            self._pending = i
            self._k += 1
            return
        cur = torch.cuda.current_stream(self.fe.device)
        with torch.cuda.stream(self.stream):
            if not input_ready:
                self.stream.wait_stream(cur)                    # `wav` was produced on the caller's stream
            if self._free[i] is not None:
                self.stream.wait_event(self._free[i])           # the step that consumed buffer i has been issued and finished
            self.fe(wav, out=self.feat[i])
            self._ready[i].record(self.stream)
        self._pending = i
        self._k += 1

    def get(self) -> torch.Tensor:
        """Features of the most recent submit, ordered after their kernel on the caller's stream.  The buffer stays valid
        until the submit after next."""
        i = self._pending
        if self.stream is None:
            return self.feat[i]
        cur = torch.cuda.current_stream(self.fe.device)
        cur.wait_event(self._ready[i])
        j = 1 - i                                               # the other buffer was consumed by the step just issued
        ev = torch.cuda.Event()
        ev.record(cur)
        self._free[j] = ev
        return self.feat[i]
