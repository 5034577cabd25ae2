// Argument blocks and host-side launchers of every kernel family; shared between the kernel
// translation units and the network engine (net.cpp).
#pragma once
#include "tcr_common.h"

namespace tcr {

// ---- conv.hip : VALU direct convolutions ----------------------------------------------------
enum ConvEpilogue { EPI_RAW = 0, EPI_AFFINE = 1 };

struct ConvArgs {
    const float* x;         // [B][Cin][Tpi]
    const float* w;         // [K][Cin][Cout]
    float* y;               // [B][Cout][Tpo]
    const float* scale;     // [Cout]  (EPI_AFFINE)
    const float* shift;     // [Cout]
    const float* res;       // [B][Cout][Tpo] residual added before the final ReLU, or nullptr
    int npos;               // B * Tout
    int cin, cout;
    int tpi, tout, tpo;
    int xoff;               // HALO - pad_lo
    int relu;
    // data-gradient use of the MFMA conv (EPI_RAW): output position t lands at row index t * ostride + ooff, and an
    // optional tensor is accumulated into the result (the block's shortcut-branch gradient)
    int ostride, ooff;      // 1, 0 for a forward conv (0 is read as 1)
    const float* add;       // [B][Cout][Tpo] (or [B][Cout] when add_bcast), or nullptr
    const float* add_mask;  // add is gated by [add_mask > 0]
    int add_bcast;
};

struct DgradArgs {
    const float* dy;        // [B][Cout][Tpo]
    const float* wt;        // [K][Cout][Cin]
    float* dx;              // [B][Cin][Tpi]
    const float* add;       // optional [B][Cin][Tpi] tensor accumulated into dx, or nullptr
    const float* add_mask;  // optional: add is gated by [add_mask > 0] (ReLU of the residual sum)
    int ngrp;               // B * U, U = ceil(Tin / S)
    int ugrp;               // U
    int cin, cout;
    int tin, tpi, tout, tpo;
    int pad_lo;
    int add_bcast;          // add is [B][Cin] broadcast over time
};

int pick_channel_tile(int c);
int launch_conv_fwd(int k, int stride, const ConvArgs& a, int epi, hipStream_t s);
int launch_conv_dgrad(int k, int stride, const DgradArgs& a, hipStream_t s);
int launch_transpose_weights(const float* w, float* wt, int k, int cin, int cout, hipStream_t s);

// ---- mfma.hip : matrix-core contractions ----------------------------------------------------
enum { MF_RAW = 0, MF_AFFINE = 1 };

// Per-channel sums taken in a conv kernel's EPILOGUE, while the output tile is still in registers (DS-CNN training: no
// separate reduction pass over the tensor):
//   forward  (raw == nullptr): q1 = y, q2 = y * y of the kernel's own output                  (chan_reduce MODE 0)
//   backward (raw != nullptr): the kernel writes dA, the gradient wrt a BN unit's activation; with raw = that unit's raw
//       conv output (same layout as the kernel's output), dz = dA * [fmaf(raw, self_scale, self_shift) > 0],
//       q1 = dz, q2 = dz * (raw - mean) * invstd                                              (chan_reduce MODE 1)
// partial: [rows][2][C] rows of per-workgroup sums (how many rows: the kernel's *_sum_rows()), added up in double, in a
// fixed order, by the finalize kernels of bn.hip.
struct EpiSums {
    float* partial = nullptr;
    const float* raw = nullptr;
    const float* mean = nullptr;
    const float* invstd = nullptr;
    const float* self_scale = nullptr;
    const float* self_shift = nullptr;
};

// BN backward of the unit whose conv the kernel differentiates, applied WHERE dy IS READ instead of by a bn_bwd_apply pass that
// writes it (units whose dy has few readers: depthwise, conv_1): with da = the gradient wrt the unit's activation, raw = its raw
// conv output,   dy = k1 * (dz - k2 - (raw - mean) * k3),   dz = da * [fmaf(raw, self_scale, self_shift) > 0]
// (bn_bwd_apply_kernel's expression).  da == nullptr: off, the kernel's dz argument is dy itself.
struct BnBwdFly {
    const float* da = nullptr;
    const float* raw = nullptr;
    const float* mean = nullptr;
    const float* k1 = nullptr;
    const float* k2 = nullptr;
    const float* k3 = nullptr;
    const float* self_scale = nullptr;
    const float* self_shift = nullptr;
};

struct Conv1x1Args {
    const float* x;         // [B][Cin][Tpi]
    const float* w;         // [Cin][Cout]
    float* y;               // [B][Cout][Tpo]
    const float* scale;
    const float* shift;
    int npos, cin, cout, tpi, tout, tpo, stride, relu;
    // LDS-tiled kernel only (DS-CNN training): x is a RAW train-mode conv output and the kernel stages relu(x * in_scale[ci] +
    // in_shift[ci]) -- the producing unit's BN + ReLU, bitwise bn_apply's expression -- so that tensor is never materialised
    const float* in_scale = nullptr;
    const float* in_shift = nullptr;
    EpiSums sums;
    // LDS-tiled kernel, data-gradient form with sums: x is the gradient wrt the pointwise unit's ACTIVATION and fly.raw that unit's raw
    // output (fly.da is not used); the kernel stages dy = BN backward of it (BnBwdFly's expression) and writes it to dy_out
    // (interior + zeroed halos) for the filter gradient -- no bn_bwd_apply pass in the backward's main chain
    BnBwdFly fly;
    float* dy_out = nullptr;
};

int launch_conv1x1(const Conv1x1Args& a, int epi, hipStream_t s);
bool conv1x1_lds_covers(int cin, int cout);                 // launch_conv1x1 takes the LDS-tiled kernel (in_scale / sums honoured)
int conv1x1_sum_rows(int npos);                             // partial rows its epilogue writes
bool pw_wgrad_lds_covers(int cin, int cout, int tp);        // launch_conv_wgrad(1, 1, ...) takes pw_wgrad_lds_kernel (x_scale / x_shift honoured)
// implicit-GEMM k x 1 conv on the matrix cores; returns 1 (nothing launched) when the shape does not fit
int launch_conv_mfma(int k, int stride, const ConvArgs& a, int epi, hipStream_t s);
// dx = dgrad of a (k, stride) conv, computed as stride-1 MFMA convs over dy with re-arranged weights (scratch `wt`);
// returns 1 when the shape is not covered
// wt_ready: `wt` already holds the re-arranged weights (launch_dgrad_weights_multi)
int launch_conv_dgrad_mfma(int k, int stride, int pad_lo, const float* w, float* wt, const float* dy, float* dx, const float* add,
                           const float* add_mask, int add_bcast, int batch, int cin, int cout, int tin, int tout, hipStream_t s,
                           bool wt_ready = false, unsigned add_phases = ~0u);      // add_phases: bit r = output phase r takes `add`
unsigned conv_dgrad_phases(int k, int stride, int pad_lo, int tin);
bool conv_dgrad_mfma_covers(int k, int stride, int cout);
int launch_conv_mfma_with_down(const ConvArgs& a, const float* w_down, float* y_down, const float* scale_down,
                               const float* shift_down, int pad_lo, int epi, hipStream_t s);
size_t wgrad_partial_floats(int k, int cin, int cout, int batch, bool fine = false);     // fine: see wgrad_chunks_for
int wgrad_chunks(int batch);
int wgrad_chunks_for(int batch, bool fine);
int launch_wgrad_reduce(const float* partial, float* dw, int nchunk, int k, int cin, int cout, int cin_pad, int cout_pad,
                        int cout_all, int co_base, hipStream_t s);
// x_scale / x_shift (pointwise LDS kernel only): x is a raw conv output, the operand is relu(x * x_scale[ci] + x_shift[ci])
int launch_conv_wgrad(int k, int stride, int pad_lo, const float* x, const float* dy, float* dw, float* scratch,
                      int batch, int cin, int cout, int tpi, int tout, int tpo, hipStream_t s,
                      const float* x_scale = nullptr, const float* x_shift = nullptr, bool x_slack = false);
// Training engines that give every layer its own partial-slab scratch launch only the split-K kernel per layer
// (launch_conv_wgrad_partial) and sum all layers' slabs in ONE launch at the end of backward (launch_wgrad_reduce_multi);
// likewise the re-arranged data-gradient weights of every layer are produced by one launch up front.
constexpr int kMultiMax = 64;        // >= 1 + 3 * TCR_MAX_BLOCKS conv layers
struct WgradReduceEntry {
    const float* partial;
    float* dw;
    int nchunk, k, cin, cout, cin_pad, cout_pad;
};
struct WgradReduceMulti {
    int n;
    WgradReduceEntry e[kMultiMax];
};
bool conv_wgrad_deferrable(int k, int cin, int cout);         // single slab (Cout <= 80), slab kernel
bool conv_wgrad_fly_covers(int k, int stride, bool x_slack);  // launch_conv_wgrad_partial honours a WgradFly for this shape
struct WgradFly;
int launch_conv_wgrad_partial(int k, int stride, int pad_lo, const float* x, const float* dy, float* scratch, int batch, int cin, int cout,
                              int tpi, int tout, int tpo, WgradReduceEntry* entry, hipStream_t s, bool fine = false, bool x_slack = false,
                              const WgradFly* fly = nullptr);
// x_slack: x and dy are followed by >= 64 readable floats (tensors inside a workspace): allows the 16-byte-load kernel
WgradReduceEntry conv_wgrad_entry(int k, int cin, int cout, int batch, const float* scratch, float* dw, bool fine = false);
int launch_wgrad_reduce_multi(const WgradReduceMulti& m, hipStream_t s);
struct DgradWeightsEntry {
    const float* w;
    float* wt;
    int k, cin, cout, stride, pad_lo;
};
struct DgradWeightsMulti {
    int n;
    DgradWeightsEntry e[kMultiMax];
};
int launch_dgrad_weights_multi(const DgradWeightsMulti& m, hipStream_t s);
struct BwdPrologueArgs {        // head backward (launch_head_bwd's arguments) + the gradient arena to clear
    const float* dlogits;
    const float* wfc;
    const float* dscale;
    float* dpool;
    int batch, c, nc;
    float* zero;
    int64_t zero_n;
};
int launch_bwd_prologue(const BwdPrologueArgs& h, const DgradWeightsMulti& m, hipStream_t s);      // head backward + arena zero fill + every layer's data-gradient weights, one launch

// ---- fused.hip : whole-network eval forward, activations resident in LDS ---------------------
constexpr int kFusedMaxLayers = 32;

struct FusedLayer {
    int k, stride, cin, cout, tin, tout, pad_lo, relu;
    int in_buf, out_buf, res_buf;       // LDS buffer ids (res_buf < 0: none)
    int w_off, ss_off, c_pad;           // float offsets into params / scale-shift table
    int no_barrier;                     // the next layer reads the same input and writes another buffer: no s_barrier in between
    // floats per utterance of this layer's input / output / shortcut rows (<= buf_sz of the buffer): the host pads the stride of a
    // layer's output so that the 32 lanes of one LDS read of its consumer fall into 32 distinct banks (net.cpp: fused_strides)
    int in_sz, out_sz, res_sz;
};

struct FusedArgs {
    const float* params;
    const float* ss;
    const float* feat;          // [B][C0][Tp0]
    float* logits;
    float* probs;
    float* ranges;              // may be nullptr
    int batch, group, n_groups, n_layers;
    int buf_off[3];             // float offset of each LDS buffer
    int buf_sz[3];              // floats per utterance in each buffer
    int in_c, in_tp;            // feature rows: channels x padded length
    int feat_buf;               // buffer holding the last block output
    int feat_c, feat_t, nc;
    int feat_sz;                // floats per utterance of the last block output
    int fc_off, fc2_off;
    int in_global;              // the first layer reads the feature rows straight from global memory (no LDS copy)
    FusedLayer layer[kFusedMaxLayers];
};

constexpr int kSmallBatchMax = 64;      // batches up to this many utterances take the small-batch (latency) kernel where it exists
// returns 1 when the launch could not be configured (caller falls back to the per-layer kernels)
int launch_net_fused(const FusedArgs& a, size_t lds_bytes, int grid, int waves, int ring, hipStream_t s);

// ---- train_fused.hip : group-resident phases of the train-mode forward ---------------------------
constexpr int kPhaseMaxRows = 1024;     // partial rows (= workgroups) a phase may write per BN layer
struct PhaseSrc {           // how a phase's input activation X [c][t] is produced while it is staged into LDS
    int kind;               // 0: plain rows `a` (features, or an already materialised activation); 1: relu(bn(a)); 2: relu(bn(a) + S)
    const float* a;         // [B][c][tp]
    const float* ss_a;      // scale at [0, c), shift at [c_pad_a, c_pad_a + c)
    int c_pad_a;
    int s_kind;             // kind 2: 0: S = rows `s` (identity shortcut); 1: S = relu(bn(s)) (the block's `down` branch)
    const float* s;
    const float* ss_s;
    int c_pad_s;
    float* out_x;           // where X is materialised for backward (halo zeroed), or nullptr
    float* out_s;           // where S is materialised (s_kind 1), or nullptr
    int c, t;
};
struct PhaseLayer {
    int k, stride, cin, cout, tin, tout, pad_lo;
    int w_off;              // floats into params
    float* raw;             // [B][cout][tpo] raw conv output (interior written)
    float* partial;         // [rows][2][cout] per-workgroup sums / sums of squares
};
struct TrainPhaseArgs {
    const float* params;
    PhaseSrc src;
    int n_layers;           // 0: staging only (materialise X / S)
    PhaseLayer layer[2];
    int batch;
    float* pool_sum = nullptr;      // closing phase (n_layers == 0): also [B][c] sums over time of the staged activation, frames in order -- the head's pooling
    int group, n_groups, in_sz, cstat, stat_off, nw; // (set by the launcher)
    int vec_stage = 0, tab_off = 0;                  // (launcher) staging with 16-byte accesses and the per-channel coefficients in an LDS table at tab_off
};
// rows_out: partial rows written per layer (= workgroups).  Returns 1 when the phase does not fit (nothing launched).
int launch_train_phase(TrainPhaseArgs a, int* rows_out, hipStream_t s);
int train_phase_rows(const TrainPhaseArgs& a);      // the same number without launching (-1: does not fit)

// ---- train_fused_bwd.hip : group-resident phases of the backward pass -----------------------------
struct BwdSrc {             // a gradient wrt a conv's raw output, built while it is staged: dy = k1 (dz - k2 - (raw - mean) k3)
    int kind;               // 0: unused; 1: from (da, masks, raw, coefficients); 2: plain rows `da` (an already materialised dy)
    const float* da;        // gradient wrt the unit's activation [B][c][tp] (or [B][c] when bcast)
    int bcast;
    const float* m1;        // ReLU mask sources (activation > 0), or nullptr
    const float* m2;
    const float* raw;
    const float* mean;
    const float* k1;
    const float* k2;
    const float* k3;
    float* out_dy;          // where dy is materialised (halo zeroed) for the weight-gradient kernels, or nullptr
    int c, t;
};
struct BwdLayer {           // data gradient of one conv, accumulated into the phase's dx rows
    int src;                // index of the staged dy
    int k, stride, pad_lo;
    int cin, tin;           // channels / frames of dx (the conv's input)
    int cout, tout;         // channels / frames of dy (the conv's output)
    const float* wt;        // re-arranged weights (launch_dgrad_weights_multi)
};
struct BwdStat {            // sums of the BN unit whose output activation dx is the gradient of
    int on;
    const float* m1;
    const float* m2;
    const float* raw;
    const float* mean;
    const float* invstd;
    float* partial;         // [rows][2][out_c]
};
struct TrainBwdPhaseArgs {
    BwdSrc src[2];
    int n_layers;           // 0: staging only (materialise dy)
    BwdLayer layer[2];
    int out_c, out_t;
    float* out_dx;          // [B][out_c][tp]
    const float* add;       // identity shortcut: dx += add [add_mask > 0]; [B][out_c][tp] or [B][out_c] (add_bcast)
    const float* add_mask;
    int add_bcast;
    BwdStat stat[2];
    int batch;
    int group, n_groups, src_sz[2], out_sz, src_off[2], out_off, cstat, stat_off, tpc, nw;      // (set by the launcher)
};
int launch_train_bwd_phase(TrainBwdPhaseArgs a, int* rows_out, hipStream_t s);       // 1: does not fit (nothing launched)
int train_bwd_phase_rows(const TrainBwdPhaseArgs& a);                                // -1: does not fit

// ---- bwd_lazy.hip : TC-ResNet backward without BN passes ("lazy" BN backward) -------------------
// The per-layer chain was, for every BN unit: reduce (sum dz, sum dz xhat) -> finalize + bn_bwd_apply (writes dy) -> data gradient
// (one launch per stride phase) + filter gradient, i.e. two elementwise passes over HBM that exist only because BN backward is a
// separate kernel.  Here dy is NEVER materialised:
//   * a unit's gradient tensor gz holds the gradient wrt its activation with every ReLU mask that does not depend on the unit
//     already applied by the kernel that wrote it (the block-output mask [out > 0]; for single-consumer units also their own mask);
//   * the data-gradient kernel stages  dy = k1 (dz - k2 - (raw - mean) k3)  into LDS for a group of whole utterances while it reads
//     gz / raw (bn_bwd_apply's expression), runs every stride phase -- and the block's 1x1 shortcut conv, whose gradient meets
//     conv_a's in the accumulators -- from that image on the matrix cores, and its EPILOGUE applies the next mask, stores the
//     next unit's gz and leaves that unit's (those units') backward sums as one partial row per workgroup;
//   * the filter-gradient kernel computes dy in registers where it loads it (WgradFly).
// Between two such kernels only the tiny finalize (partial rows -> dgamma, dbeta, k1..k3) runs.
struct LazySrc {            // dy of one BN unit, built while it is staged
    const float* gz;        // [B][c][t + 2 halo] gradient wrt the unit's activation (see above)
    const float* raw;       // [B][c][t + 2 halo] the unit's raw conv output
    const float* k1;        // per-channel coefficients left by bn_bwd_finalize
    const float* k2;
    const float* k3;
    const float* mean;
    const float* self_scale;    // the unit's own ReLU mask still to be applied: dz = gz [fmaf(raw, scale, shift) > 0]; or nullptr
    const float* self_shift;
    const float* tab;       // the same coefficients packed per channel by bn_bwd_finalize: [c][8] = k1, k2, k3, mean, scale, shift (0, 1 without own mask)
    int c, t;
};
struct LazyLayer {          // data gradient of one conv (k x 1, stride 1 or 2) whose dy is src
    int src;
    int k, stride, pad_lo;
    const float* wt;        // phase-major re-arranged weights (bwd_prologue / dgrad_weights_body)
};
struct LazyStat {           // backward sums of a unit whose activation gradient this kernel writes
    int on;
    const float* raw;
    const float* mean;
    const float* invstd;
    const float* self_scale;    // the unit's own ReLU mask (nullptr: none -- a block's conv_b)
    const float* self_shift;
    float* partial;             // [rows][2][out_c]
};
struct BwdLazyArgs {
    LazySrc src[2];
    int n_layers;           // 1: conv_b, or conv_a of an identity block; 2: conv_a + the block's shortcut conv (same stride)
    LazyLayer layer[2];
    int out_c, out_t;       // channels / frames of the gradient written (the convs' input)
    float* out_g;           // [B][out_c][out_t + 2 halo]
    const float* add;       // identity shortcut: + the block-output gradient (same shape), or nullptr
    const float* mask_act;  // block-output activation the written gradient is masked with ([mask_act > 0]), or nullptr
    int store_self;         // the stored gradient also carries stat[0]'s own mask (its only consumer is that unit)
    LazyStat stat[2];
    int batch;
    // (set by the launcher)
    int group, n_groups, ks, mt, nw, qmax, taps;
    int vec_stage;          // staging with 16-byte loads / LDS stores (rows a multiple of four floats per utterance, aligned tensors)
    int img_off[2], red_off, stat_off, coef_off, ecoef_off, cstat;
    int nu[2];              // output positions per utterance of stride phase 0 / 1
    int cnt[2][2], dmin[2][2], wbase[2][2];     // [layer][phase]: taps, first dy offset, float offset of the phase's weights in wt
    long long* dbg;         // (TCR_DEBUG_LAZY_TS: per-workgroup cycle stamps of the kernel's sections; nullptr otherwise)
};
int launch_bwd_lazy(BwdLazyArgs a, int* rows_out, hipStream_t s);      // 1: shape not covered (nothing launched)
int bwd_lazy_rows(const BwdLazyArgs& a);                               // partial rows it writes (-1: not covered)

// dy computed where the filter-gradient kernel loads it (raw == nullptr: off, `dy` is read as is)
struct WgradFly {
    const float* raw = nullptr;
    const float* k1 = nullptr;
    const float* k2 = nullptr;
    const float* k3 = nullptr;
    const float* mean = nullptr;
    const float* self_scale = nullptr;
    const float* self_shift = nullptr;
};

// ---- bn.hip ---------------------------------------------------------------------------------
constexpr int kBnMaxLayers = 40;

struct BnFoldArgs {
    const float* params;
    const float* stats;
    float* out;                 // scale at out_off, shift at out_off + c_pad
    int n;
    float eps;
    int c[kBnMaxLayers];
    int c_pad[kBnMaxLayers];
    int64_t gamma_off[kBnMaxLayers], beta_off[kBnMaxLayers], mean_off[kBnMaxLayers], var_off[kBnMaxLayers], out_off[kBnMaxLayers];
    int64_t bias_off[kBnMaxLayers];     // conv bias folded into the shift (DS-CNN), or -1; gamma_off -1 = no scale
};

struct ChanReduceArgs {
    const float* y;         // [B][C][Tp]
    const float* da;        // MODE 1: gradient wrt the activation ([B][C][Tp], or [B][C] when bcast)
    const float* m1;        // MODE 1: optional ReLU mask sources (activation > 0)
    const float* m2;
    const float* mean;      // MODE 1
    const float* invstd;    // MODE 1
    float* partial;         // [nchunk][2][C]
    int npos, c, t, tp;
    int pos_per_block;
    int bcast;              // da is [B][C] broadcast over time
    // MODE 1: the unit's OWN ReLU mask recomputed from y instead of read from its activation: [fmaf(y, scale, shift) > 0] -- the
    // expression the forward's bn_apply / staging uses, so the mask is bitwise the activation's (one tensor read less per pass)
    const float* self_scale;
    const float* self_shift;
    float* g_out;           // MODE 1 (scalar kernel): also store dz at the element's place ([B][C][Tp] interior) -- the lazy backward's last block,
                            // whose pooled, broadcast gradient becomes a masked tensor on the way; or nullptr
    // MODE 2 (scalar kernel; the lazy backward's last block, round 6): a SECOND unit's sums in the same pass -- the block's shortcut
    // unit takes the same dz under its own mask [fmaf(y2, self_scale2, self_shift2) > 0] with xhat from y2: what a second launch
    // reading g_out back computes, one launch earlier in the step's dependency chain
    const float* y2;
    const float* mean2;
    const float* invstd2;
    const float* self_scale2;
    const float* self_shift2;
    float* partial2;
};

struct BnFinalizeArgs {
    const float* partial;       // [nchunk][2][C] per-workgroup partial sums (nchunk > 0), reduced here in double ...
    int nchunk;                 // ... or 0: take the already reduced (possibly cross-replica) `sums`
    const double* sums;         // [2][C]: sum y, sum y^2
    const float* gamma;
    const float* beta;
    float* moving_mean;
    float* moving_var;
    float* scale;               // out: gamma * invstd
    float* shift;               // out: beta - mean * scale
    float* mean;                // out (saved for backward)
    float* invstd;              // out (saved for backward)
    int c;
    double count;
    float decay, eps;
    int cbw = 0;                // (set by the launcher: channels per workgroup, bn_finalize_cb(nchunk))
};

struct BnApplyArgs {
    const float* y;
    const float* scale;
    const float* shift;
    const float* res;
    float* out;
    int64_t total;      // B * C * Tp
    int c, t, tp, relu;
    float inv_tp;       // (set by the launcher)
};

struct BnBwdFinalizeArgs {
    const float* partial;   // as in BnFinalizeArgs
    int nchunk;
    const double* sums;     // [2][C]: sum dz, sum dz*xhat
    const float* gamma;
    const float* invstd;
    float* dgamma;
    float* dbeta;
    float* k1;
    float* k2;
    float* k3;
    int c;
    double count;           // elements per channel over the batch the statistics span
    float grad_scale;       // sync BN: sums are global on every replica, and the arena all-reduce will add the
                            // replicas' copies up again -> store dgamma/dbeta divided by the replica count
    int cbw = 0;            // (set by the launcher)
    // lazy backward: also one packed row per channel, tab[c][8] = k1, k2, k3, mean, own-mask scale, shift (0, 1 when self_scale is null), 0, 0
    float* tab = nullptr;
    const float* mean = nullptr;
    const float* self_scale = nullptr;
    const float* self_shift = nullptr;
};

struct BnBwdApplyArgs {
    const float* y;
    const float* da;
    const float* m1;
    const float* m2;
    const float* mean;
    const float* k1;
    const float* k2;
    const float* k3;
    float* dy;
    int64_t total;
    int c, t, tp, bcast;
    float inv_tp;           // (set by the launcher)
    int accumulate;         // dy += ... (interior positions only) instead of dy = ...: gradient buffers with several producers (net2d)
    const float* self_scale = nullptr;      // as in ChanReduceArgs
    const float* self_shift = nullptr;
};

int launch_bn_fold(const BnFoldArgs& a, hipStream_t s);
int chan_reduce_chunks(int npos);
int chan_reduce_launch_chunks(int npos, int t);      // t: frames per utterance (chunks are whole utterances where one fits)
int launch_chan_reduce(int mode, ChanReduceArgs a, int* nchunk_out, hipStream_t s);
int launch_chan_sums(const float* partial, int nchunk, int c, double* sums, hipStream_t s);
int bn_finalize_cb(int nchunk);         // channels per finalize workgroup for nchunk partial rows
int launch_bn_finalize(const BnFinalizeArgs& a, hipStream_t s);
int launch_bn_finalize2(const BnFinalizeArgs& a, const BnFinalizeArgs& b, hipStream_t s);       // two units, one launch (bitwise two launch_bn_finalize calls)
int launch_bn_apply(const BnApplyArgs& a, hipStream_t s);
int launch_bn_bwd_finalize(const BnBwdFinalizeArgs& a, hipStream_t s);
int launch_bn_bwd_finalize2(const BnBwdFinalizeArgs& a, const BnBwdFinalizeArgs& b, hipStream_t s);    // two units, one launch (bitwise two launch_bn_bwd_finalize calls)
int launch_bn_bwd_apply(const BnBwdApplyArgs& a, hipStream_t s);
int launch_bn_bwd_apply_fused(const BnBwdFinalizeArgs& f, const BnBwdApplyArgs& a, hipStream_t s);     // 1: not applicable, launch the pair

// ---- head.hip -------------------------------------------------------------------------------
struct HeadArgs {
    const float* feat;          // [B][C][Tp]
    const float* wfc;           // [C][NC]
    const float* wfc2;          // [C][2]
    const float* labels;        // [B][NC] one-hot (train)
    float* logits;              // [B][NC]
    float* probs;               // [B][NC]
    float* ranges;              // [B][2] or nullptr
    float* dropped;             // [B][C]   (train) input of fc after dropout
    float* dscale;              // [B][C]   (train) d(dropped)/d(sum over time) = mask / keep_prob / T
    float* dlogits;             // [B][NC]  (train) (p - y) / global_batch
    float* loss_utt;            // [B]      (train) -sum_k y_k log p_k
    const float* bias;          // [NC] added to the logits (slim.fully_connected of DS-CNN), or nullptr
    int batch, c, nc, t, tp;
    float keep_prob;
    uint64_t seed;
    int64_t sample_offset;
    float inv_global_batch;
    float label_smoothing;
    float pool_scale;           // (train) overrides 1/t in dscale when feat is an already pooled [B][C] map (DS-CNN), or 0
    const float* pool_sum = nullptr;    // [B][C] sums over time of feat, frames in order (the closing training phase leaves them): the head skips its walk over the rows
};

int launch_head_fwd(const HeadArgs& a, bool train, hipStream_t s);
int launch_head_bwd(const float* dlogits, const float* wfc, const float* dscale, float* dpool, int batch, int c, int nc, hipStream_t s);
int fc_wgrad_chunks(int batch);
int launch_fc_wgrad(const float* dropped, const float* dlogits, float* partial, float* dw, int batch, int c, int nc, hipStream_t s);
int launch_sum_vector(const float* in, int n, float* out, hipStream_t s);
int launch_bias_grad(const float* dlogits, int batch, int nc, float* db, hipStream_t s);

// ---- dscnn_bwd.hip : backward kernels of the depthwise-separable baseline ----------------------
struct DsDwBwdArgs {
    const float* dz;        // [B][C][Ppo] gradient wrt the depthwise conv output
    const float* w;         // [3][3][C][1]
    float* dx;              // [B][C][Ppi]
    int64_t planes;         // B * C
    int c, h_in, w_in, ppi, oh, ow, ppo, sh, sw, pad_t, pad_l;
    EpiSums sums;           // backward sums of the unit whose activation gradient dx is (LDS kernel only; rows = B: [n][2][C])
    BnBwdFly fly;           // (LDS kernel only)
};
int launch_dscnn_dw_dgrad(const DsDwBwdArgs& a, hipStream_t s);
bool dscnn_dw_dgrad_lds_covers(int h_in, int w_in, int pad_t, int pad_l);

struct DsDwWgradArgs {
    const float* x;         // [B][C][Ppi] input of the depthwise conv
    const float* dz;        // [B][C][Ppo]
    float* partial;         // [nchunk][9][C]
    int batch, c, h_in, w_in, ppi, oh, ow, ppo, sh, sw, pad_t, pad_l;
    int utt_per_block;
    const float* x_scale = nullptr;     // x is a raw conv output: the operand is relu(x * x_scale[c] + x_shift[c])
    const float* x_shift = nullptr;
    BnBwdFly fly;
};
size_t dscnn_dw_wgrad_partial_floats(int batch, int c);
int launch_dscnn_dw_wgrad(DsDwWgradArgs a, float* dw, hipStream_t s);

struct DsConv1WgradArgs {
    const float* feat;      // [B][W_in][Tp_in] front-end planar features
    const float* dz;        // [B][Cout][Pp]
    float* partial;         // [nchunk][taps_pad][Cout_pad]
    int batch, cout, cout_pad, taps, taps_pad;
    int h_in, w_in, tp_in, oh, ow, pp, kh, sh, sw, pad_t, pad_l;
    int utt_per_block;
    BnBwdFly fly;
};
size_t dscnn_conv1_wgrad_partial_floats(int batch, int kh, int cout);
int launch_dscnn_conv1_wgrad(DsConv1WgradArgs a, float* dw, hipStream_t s);

}  // namespace tcr
