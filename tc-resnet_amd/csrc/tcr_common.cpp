// Error plumbing and ABI bookkeeping shared by every translation unit.
#include "tcr_common.h"

#include <cstdlib>
#include <cstring>
#include <initializer_list>
#include <mutex>

namespace tcr {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return TCR_ERR_HIP;
    }
    return TCR_OK;
}

// The library's internal streams: ONE set per device for the whole process, chosen on first use and kept until exit.
// HIP multiplexes streams onto a handful of hardware queues (four by default), and two streams that land on one queue
// serialise.  With a pair of streams per net object, the 4th and 6th TCResNet8 created in a process trained 33 % slower
// than the first (1355 vs 1013 us per step, scripts/stream_alias_check.py); with one fixed pair, how many streams OTHER code
// had created first decided it (five foreign streams: 1370 us); the two-stream inference pipeline runs at 272 us per batch on
// streams with queues of their own, 313 on two streams that share one, 322 when the network's shares the caller's
// (scripts/stream_concurrency_probe.py).  So the set is CHOSEN, not just created: candidates are created and probed --
// a 2 ms spin on one stream, an empty kernel on the other: the empty kernel finishing first means different queues --
// until streams 0, 1, 2 and the first caller's stream are pairwise concurrent (four queues), and stream 3 is concurrent with
// the caller's and with stream 2; the other candidates are destroyed.  One-off, ~20-60 ms.  Streams of the lowest / highest
// PRIORITY (hoping for a queue of another level, never the caller's) were worse: 2112 us per step with two foreign streams.
// Roles: 0 filter gradients / shortcut branch, 1 classifier gradients, 2 front-end of the next batch, 3 network of the
// two-stream inference pipeline (2 and 3 are handed to the host side by tcr_internal_stream).
__global__ void stream_probe_spin_kernel(long long ticks) {
    const long long t0 = wall_clock64();            // 100 MHz constant clock
    while (wall_clock64() - t0 < ticks) {}
}
__global__ void stream_probe_empty_kernel() {}

// true: work on `b` overtakes work on `a` (different hardware queues)
static bool streams_concurrent(hipStream_t a, hipStream_t b, hipEvent_t ea, hipEvent_t eb) {
    if (hipStreamSynchronize(a) != hipSuccess || hipStreamSynchronize(b) != hipSuccess) return false;
    hipLaunchKernelGGL(stream_probe_spin_kernel, dim3(1), dim3(1), 0, a, 200000LL);     // 2 ms
    if (hipEventRecord(ea, a) != hipSuccess) return false;
    hipLaunchKernelGGL(stream_probe_empty_kernel, dim3(1), dim3(1), 0, b);
    if (hipEventRecord(eb, b) != hipSuccess || hipEventSynchronize(eb) != hipSuccess) { (void)hipEventSynchronize(ea); return false; }
    const bool overtook = hipEventQuery(ea) == hipErrorNotReady;
    (void)hipEventSynchronize(ea);
    return overtook;
}

constexpr int kSharedStreams = 4;

// The set is chosen ONCE per device and process, against the stream of the FIRST call that needs an internal stream (training step,
// pipeline, or an explicit tcr_internal_stream(i, stream) at start-up, which is the way to pay the one-off probe cost -- tens of ms of
// 2 ms spin kernels and stream synchronisations on that stream -- outside a latency-sensitive call).  The handles are handed to callers,
// so the set is never re-chosen: when the first call arrives on a stream under graph capture the probe is skipped (synchronising would
// invalidate the capture) and the first unused candidates are taken as they come -- warm the library up before capturing.
static void choose_streams(hipStream_t caller, hipStream_t (&out)[kSharedStreams]) {
    constexpr int kCand = 12;
    hipStream_t cand[kCand] = {};
    bool used[kCand] = {};
    int nc = 0;
    for (; nc < kCand; ++nc)
        if (hipStreamCreateWithFlags(&cand[nc], hipStreamNonBlocking) != hipSuccess) break;
    hipEvent_t ea = nullptr, eb = nullptr;
    bool probe = hipEventCreateWithFlags(&ea, hipEventDisableTiming) == hipSuccess && hipEventCreateWithFlags(&eb, hipEventDisableTiming) == hipSuccess;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;       // a caller's stream under graph capture must not be synchronised: no probing then
    if (hipStreamIsCapturing(caller, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) probe = false;
    auto pick = [&](std::initializer_list<hipStream_t> apart) -> hipStream_t {
        for (int pass = 0; pass < 2; ++pass)                    // pass 1: nothing qualified (or no probing possible) -- the first unused candidate
            for (int i = 0; i < nc; ++i) {
                if (used[i]) continue;
                bool ok = true;
                if (pass == 0) {
                    ok = probe;
                    for (hipStream_t o : apart) ok = ok && (o == nullptr || streams_concurrent(o, cand[i], ea, eb));    // (a slot that could not be filled constrains nothing)
                }
                if (ok) { used[i] = true; return cand[i]; }
            }
        return nullptr;
    };
    out[0] = pick({caller});
    out[1] = pick({caller, out[0]});
    out[2] = pick({caller, out[0], out[1]});
    out[3] = pick({caller, out[2]});
    for (int i = 0; i < nc; ++i) if (!used[i]) (void)hipStreamDestroy(cand[i]);
    if (ea) (void)hipEventDestroy(ea);
    if (eb) (void)hipEventDestroy(eb);
    (void)hipGetLastError();
}

// The fork / join events order streams of one device; no host thread ever waits on them.  A default event's record carries a SYSTEM-scope
// release (cache write-back for host / peer visibility) that such an edge does not need: TCR_EVENT_FENCE=1 keeps it (A/B).
unsigned internal_event_flags() {
    static const bool fence = getenv("TCR_EVENT_FENCE") != nullptr && getenv("TCR_EVENT_FENCE")[0] == '1';
    return fence ? hipEventDisableTiming : (hipEventDisableTiming | hipEventDisableSystemFence);
}

hipStream_t shared_stream(int idx, hipStream_t caller) {
    constexpr int kDev = 64;
    static std::mutex mu;
    static hipStream_t pool[kDev][kSharedStreams] = {};
    static bool chosen[kDev] = {};
    int dev = 0;
    if (idx < 0 || idx >= kSharedStreams || hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kDev) return nullptr;
    std::lock_guard<std::mutex> lock(mu);
    if (!chosen[dev]) { choose_streams(caller, pool[dev]); chosen[dev] = true; }
    return pool[dev][idx];
}

static int g_tune[TCR_TUNE_COUNT] = {0};

int tune_get(int knob) { return (knob >= 0 && knob < TCR_TUNE_COUNT) ? g_tune[knob] : 0; }

int device_cus() {
    static int cached[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (cached[dev] == 0) {
        int n = 0;
        cached[dev] = (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) ? n : 256;
    }
    return cached[dev];
}

}  // namespace tcr

extern "C" int tcr_tune(int knob, int value) {
    if (knob < 0 || knob >= TCR_TUNE_COUNT) {
        tcr::set_error("tcr_tune: unknown knob %d", knob);
        return TCR_ERR_ARG;
    }
    tcr::g_tune[knob] = value;
    return TCR_OK;
}

extern "C" void* tcr_internal_stream(int idx, void* caller_stream) {
    hipStream_t st = tcr::shared_stream(idx, static_cast<hipStream_t>(caller_stream));
    if (!st) tcr::set_error("tcr_internal_stream: no stream %d", idx);
    return st;
}

extern "C" int tcr_abi_version(void) { return TCR_ABI_VERSION; }
extern "C" const char* tcr_last_error(void) { return tcr::g_err; }

extern "C" const char* tcr_kernel_name(int index) {
    static const char* names[] = {
        "frontend_pk_kernel", "net_fused_kernel", "augment_kernel", "frontend_kernel", "conv_fwd_kernel", "conv_mfma_kernel", "conv1x1_mfma_kernel", "head_fwd_kernel",
        "bn_finalize_kernel", "bn_apply_kernel", "head_bwd_kernel", "bn_bwd_reduce_kernel",
        "bn_bwd_apply_kernel", "bn_bwd_apply4_kernel", "chan_reduce4_kernel", "bn_apply4_kernel", "bn_bwd_apply_fused_kernel", "conv_dgrad_kernel", "conv_wgrad_mfma_kernel", "wgrad_reduce_kernel",
        "sgd_momentum_kernel", "adam_kernel", "rmsprop_kernel", "ema_kernel", "l2_loss_kernel", "pw_wgrad_lds_kernel", "dscnn_conv1_kernel",
        "dscnn_depthwise_kernel", "dscnn_dw_dgrad_kernel", "dscnn_dw_wgrad_kernel", "dscnn_conv1_wgrad_kernel", "train_phase_kernel", "train_bwd_phase_kernel", "conv2d_mfma_kernel", "conv2d_wgrad_kernel", "pool2d_fwd_kernel",
        "pool2d_bwd_kernel", "eltwise2d_kernel", "head2d_kernel", "chan_sum2d_kernel", "features_to_plane_kernel",
        "net_fused_tc8_kernel", "net_fused_tc14w_kernel", "train_phase_s_kernel", "bwd_lazy_kernel", "conv_wgrad_mfma4_kernel", "conv_wgrad_lds_kernel",
        "conv1x1_lds_kernel", "bn_bwd_finalize_kernel", "bn_bwd_finalize2_kernel", "wgrad_reduce_multi_kernel", "frontend_pk3_kernel", "pw_wgrad_glds_kernel", "pw_wgrad_lds_p_kernel", "frontend_deploy_f64_kernel", "net_small_tc8_kernel", "bn_bwd_apply4x_kernel", "dscnn_dw_dgrad_rows_kernel", "dscnn_dw_dgrad_rows_s2_kernel", "dscnn_dw_wgrad_rows_kernel", "dscnn_dw_wgrad_rows_s2_kernel", "dscnn_depthwise_rows_kernel", "dscnn_depthwise_rows_s2_kernel", "plane_mean_block_kernel", "dscnn_conv1_dw_loop_kernel",
    };
    const int n = (int)(sizeof(names) / sizeof(names[0]));
    return (index >= 0 && index < n) ? names[index] : nullptr;
}
