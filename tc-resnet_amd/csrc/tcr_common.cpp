// Error plumbing and ABI bookkeeping shared by every translation unit.
#include "tcr_common.h"

#include <cstring>
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

// The library's internal streams: ONE set per device for the whole process, created on first use and kept until exit.
// HIP multiplexes streams onto a handful of hardware queues, and two streams that land on one queue serialise: with a
// pair of streams per net object, the 4th and 6th TCResNet8 created in a process trained 33 % slower than the first
// (1355 vs 1013 us per step, scripts/stream_alias_check.py) -- which streams shared a queue depended on how many nets
// had come and gone.  A fixed set keeps the mapping the first net got; nets driven concurrently from different caller
// streams share it (their event dependencies stay per net: correct, at worst serialised).
hipStream_t shared_stream(int idx) {
    constexpr int kDev = 64, kN = 2;
    static std::mutex mu;
    static hipStream_t pool[kDev][kN] = {};
    int dev = 0;
    if (idx < 0 || idx >= kN || hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kDev) return nullptr;
    std::lock_guard<std::mutex> lock(mu);
    // (Normal priority.  Streams of the lowest / highest priority were tried -- the idea: another priority level, another set of
    // hardware queues, never the caller's -- and were WORSE whenever other streams existed: 2112 us per TCResNet8 step with two
    // foreign streams created first, against 1035 at normal priority; scripts/stream_alias_check.py.)
    if (!pool[dev][idx] && hipStreamCreateWithFlags(&pool[dev][idx], hipStreamNonBlocking) != hipSuccess) pool[dev][idx] = nullptr;
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
    };
    const int n = (int)(sizeof(names) / sizeof(names[0]));
    return (index >= 0 && index < n) ? names[index] : nullptr;
}
