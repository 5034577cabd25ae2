/*
 * tcresnet_hip.h -- C ABI of the MI355X-native (gfx950) TC-ResNet keyword-spotting hot path.
 *
 * The reference (hyperconnect/TC-ResNet, TF 1.13 graph code) has no FFI/plugin boundary for this
 * path: every number is produced by TensorFlow ops reached from a handful of Python call sites.
 * Each entry point below replaces the TF ops behind one of those call sites (cited per function,
 * paths relative to the reference tree).  The Python host package (tc-resnet_amd/) binds this
 * header with ctypes and exposes the reference's own class/argument names on top of it.
 *
 * Conventions
 *   - every function returns 0 on success, a negative tcr_status otherwise; tcr_last_error()
 *     returns a thread-local message.  Nothing throws, nothing allocates device memory:
 *     all device buffers (including workspaces) are caller-owned, contiguous float32.
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued on it, asynchronously.
 *   - activations are planar per utterance with a zero halo on the time axis:
 *       act[b][c][TCR_HALO + t],  row length Tp = T + 2*TCR_HALO  (tcr_padded_len(T)).
 *     The halo implements TF "SAME" zero padding without per-tap bounds checks.
 *   - conv weights keep the reference checkpoint layout [k][Cin][Cout] (= TF HWIO [k,1,Cin,Cout]).
 */
#ifndef TCRESNET_HIP_H
#define TCRESNET_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TCR_ABI_VERSION 3
#define TCR_HALO 4
#define TCR_MAX_BLOCKS 16

typedef enum tcr_status {
    TCR_OK = 0,
    TCR_ERR_ARG = -1,          /* invalid argument / unsupported configuration */
    TCR_ERR_HIP = -2,          /* a HIP call or kernel launch failed */
    TCR_ERR_WORKSPACE = -3     /* workspace too small */
} tcr_status;

int tcr_abi_version(void);
const char* tcr_last_error(void);
static inline int tcr_padded_len(int t) { return t + 2 * TCR_HALO; }

/* ------------------------------------------------------------------------------------------ */
/* Front-end: MFCC / log-mel (datasets/preprocessors.py:54-96,183-194; window/stride samples   */
/* from factory/audio_nets.py:62-64; flags from datasets/audio_data_wrapper.py:61-110).        */
/* ------------------------------------------------------------------------------------------ */
typedef struct tcr_frontend_cfg {
    int32_t sample_rate;        /* --sample_rate 16000 */
    int32_t n_samples;          /* sample_rate * clip_duration_ms / 1000 */
    int32_t win;                /* window_size_samples  */
    int32_t hop;                /* window_stride_samples */
    int32_t nfft;               /* out: enclosing power of two of win (tf.contrib.signal.stft) */
    int32_t n_frames;           /* out: 1 + (n_samples - win) / hop (signal.frame, pad_end=False) */
    int32_t n_mel;              /* --num_mel_bins (64) */
    int32_t n_coef;             /* --num_mfccs for mfcc; ignored (== n_mel) for log-mel */
    float lower_hz;             /* --lower_edge_hertz 80 */
    float upper_hz;             /* --upper_edge_hertz 7600 */
    int32_t method;             /* 0 = mfcc (power spectrum + DCT-II), 1 = log_mel_spectrogram (magnitude),
                                 * 2 = deploy-path mfcc: contrib_audio.audio_spectrogram + contrib_audio.mfcc op semantics
                                 *     (datasets/preprocessors.py:98-124,196-203): magnitude-weighted mel filterbank of the op,
                                 *     log(max(x, 1e-12)), the same sqrt(2/N) DCT-II */
} tcr_frontend_cfg;

/* Fills nfft / n_frames and validates the configuration. */
int tcr_frontend_resolve(tcr_frontend_cfg* cfg);
/* Size in bytes of the constant tables (Hann window, FFT twiddles, HTK mel slopes, DCT-II). */
size_t tcr_frontend_plan_bytes(const tcr_frontend_cfg* cfg);
/* Builds the tables in HOST memory (computed in float64, stored float32); the caller uploads
 * them once to a device buffer of the same size and passes that pointer to tcr_frontend_fwd. */
int tcr_frontend_plan_init(const tcr_frontend_cfg* cfg, void* host_plan);
/* Dense [n_bins][n_mel] mel matrix / [n_mel][n_coef] DCT matrix reconstructed from a plan
 * (host side; for tests and for exporting the constants). */
int tcr_frontend_plan_mel_matrix(const tcr_frontend_cfg* cfg, const void* host_plan, float* out);
int tcr_frontend_plan_dct_matrix(const tcr_frontend_cfg* cfg, const void* host_plan, float* out);

/* wav [batch][n_samples] -> feat [batch][n_coef][Tp] (halo zeroed), Tp = tcr_padded_len(n_frames).
 * Replaces tf.contrib.signal.stft / linear_to_mel_weight_matrix / tensordot / log /
 * mfccs_from_log_mel_spectrograms (datasets/preprocessors.py:68-94,191-193). */
int tcr_frontend_fwd(const tcr_frontend_cfg* cfg, const void* plan_dev, const float* wav, int batch,
                     float* feat, void* stream);
/* The same with a per-call launch hint: `rounds` > 0 fixes the packed kernels' rounds of frames per persistent-workgroup chunk
 * (clamped to what the kernel supports; 0 = the launcher's cost model, i.e. tcr_frontend_fwd).  A caller that runs the front-end
 * next to other kernels (tcresnet_amd.pipeline) passes its own measured choice here instead of flipping the process-wide
 * TCR_TUNE_FRONTEND knob around the launch.  Results do not depend on it. */
int tcr_frontend_fwd_rounds(const tcr_frontend_cfg* cfg, const void* plan_dev, const float* wav, int batch,
                            float* feat, int rounds, void* stream);

/* "no_preprocessing" (datasets/preprocessors.py:45-49): re-layout a reference-shaped feature
 * tensor [batch][T][F] into the planar halo layout [batch][F][Tp], and back. */
int tcr_features_to_planar(const float* ntf, int batch, int t, int f, float* planar, void* stream);
int tcr_features_from_planar(const float* planar, int batch, int t, int f, float* ntf, void* stream);

/* ------------------------------------------------------------------------------------------ */
/* Network: TC-ResNet (audio_nets/tc_resnet.py:6-70, arg scope :102-123)                       */
/* ------------------------------------------------------------------------------------------ */
typedef struct tcr_tcresnet_cfg {
    char scope[32];                 /* "TCResNet8" / "TCResNet14": TF variable scope (tc_resnet.py:57,65) */
    int32_t in_channels;            /* MFCC coefficients (become channels, tc_resnet.py:17) */
    int32_t t_in;                   /* number of frames L */
    int32_t num_classes;
    int32_t n_blocks;
    int32_t channels[TCR_MAX_BLOCKS + 1];   /* n_channels after width_multiplier (tc_resnet.py:59-60) */
    float bn_decay;                 /* 0.997 (tc_resnet.py:107) */
    float bn_eps;                   /* 0.001 (slim.batch_norm default) */
} tcr_tcresnet_cfg;

typedef struct tcr_net tcr_net;     /* opaque, host-only (no device allocations) */

int tcr_tcresnet_create(const tcr_tcresnet_cfg* cfg, tcr_net** out);
void tcr_net_destroy(tcr_net* net);

/* Parameter arena layout.  Trainables live in ONE flat float32 arena (so that the optimiser step
 * and the data-parallel gradient all-reduce are one call each): all conv/fc weights first
 * (these are the L2-regularised variables of factory/audio_nets.py:175-180), then BN gamma/beta.
 * Moving statistics live in a second arena.  Offsets are in floats. */
typedef enum tcr_tensor_kind {
    TCR_WEIGHT = 0, TCR_GAMMA = 1, TCR_BETA = 2, TCR_MOVING_MEAN = 3, TCR_MOVING_VAR = 4
} tcr_tensor_kind;

typedef struct tcr_tensor_info {
    char name[96];          /* TF variable name, e.g. "TCResNet8/block0/conv0_0/BatchNorm/gamma" */
    int32_t kind;           /* tcr_tensor_kind */
    int32_t arena;          /* 0 = trainable arena, 1 = moving-stat arena */
    int64_t offset;         /* floats */
    int64_t size;           /* floats */
    int32_t shape[4];       /* TF shape, e.g. [9,1,16,24] */
    int32_t rank;
} tcr_tensor_info;

int64_t tcr_net_param_floats(const tcr_net* net);      /* trainable arena size (padded) */
int64_t tcr_net_decay_floats(const tcr_net* net);      /* prefix of the arena that is L2-regularised */
int64_t tcr_net_stat_floats(const tcr_net* net);       /* moving-stat arena size */
int tcr_net_num_tensors(const tcr_net* net);
int tcr_net_tensor_info(const tcr_net* net, int index, tcr_tensor_info* out);
int tcr_net_out_frames(const tcr_net* net);            /* L' after the last block */
int tcr_net_feat_channels(const tcr_net* net);         /* channels into fc */

/* Workspace size for a batch; train != 0 includes saved activations and gradient scratch. */
size_t tcr_net_workspace_bytes(const tcr_net* net, int batch, int train);

/* Eval-mode forward (is_training=False: BN uses moving stats, dropout off):
 * tc_resnet() + slim.softmax (audio_nets/tc_resnet.py:6-54, factory/audio_nets.py:147-156),
 * i.e. what Base.run_inference fetches (helper/base.py:86-104).
 * feat [batch][Cin][Tp]; logits/probs [batch][num_classes]; ranges [batch][2] (may be NULL). */
int tcr_net_forward_infer(const tcr_net* net, const float* params, const float* stats, const float* feat,
                          int batch, void* workspace, size_t workspace_bytes,
                          float* logits, float* probs, float* ranges, void* stream);

/* Deployable ("frozen") form of the network (factory/audio_nets.py:87-125 build_deployable_model + freeze.py:16-49
 * convert_variables_to_constants): eval-mode BN folded into per-channel (scale, shift) constants.
 *   tcr_net_frozen_floats   size of the constant table;
 *   tcr_net_fold_bn         params + moving stats -> the table (what a frozen export stores next to the conv weights);
 *   tcr_net_forward_frozen  eval forward from (conv weights in `params`, table): no variable is read, the BN entries of
 *                           `params` and the moving statistics are not needed.  Bitwise tcr_net_forward_infer. */
int64_t tcr_net_frozen_floats(const tcr_net* net);
int tcr_net_fold_bn(const tcr_net* net, const float* params, const float* stats, float* frozen_ss, void* stream);
int tcr_net_forward_frozen(const tcr_net* net, const float* params, const float* frozen_ss, const float* feat,
                           int batch, void* workspace, size_t workspace_bytes,
                           float* logits, float* probs, float* ranges, void* stream);

/* The whole eval path of one batch in ONE call: waveform -> MFCC -> (BN fold when `refold` != 0) -> network -> softmax, i.e.
 * model.build(wavs, labels, is_training=False) + session.run(outputs) of the reference (factory/audio_nets.py:41-60,
 * datasets/preprocessors.py:183-194, audio_nets/tc_resnet.py:6-54, helper/base.py:86-104).  Same kernels and results as
 * tcr_frontend_fwd + tcr_net_fold_bn + tcr_net_forward_frozen; it exists for small batches, where three host calls cost more than
 * the kernels (B <= 64: ~70 us -> the kernels' own time).
 * feat: caller-owned scratch for the features, [batch][n_coef][n_frames + 2 * TCR_HALO] floats; frozen_ss: the folded table
 * (tcr_net_frozen_floats), rewritten when `refold` is non-zero (stats may be NULL otherwise). */
int tcr_forward_waveform(const tcr_frontend_cfg* cfg, const void* plan_dev, const tcr_net* net, const float* params,
                         const float* stats, float* frozen_ss, int refold, const float* wav, int batch, float* feat,
                         void* workspace, size_t workspace_bytes, float* logits, float* probs, float* ranges, void* stream);

/* Train-mode forward (is_training=True): batch-statistics BN with moving-stat update
 * (decay, Bessel-corrected variance), inverted dropout after the global pool, softmax,
 * mean cross-entropy (factory/audio_nets.py:161-173).  Saves what backward needs in `workspace`.
 *   labels      [batch][num_classes] one-hot float (datasets/audio_data_wrapper.py:114-118)
 *   keep_prob   --dropout_keep_prob; the mask for sample i, channel c is a pure function of
 *               (seed, sample_offset + i, c) so that a sharded batch draws the same mask
 *   loss_out    device float[2]: {sum over the batch of -sum_k y log softmax, unused}
 *   global_batch  divisor of the mean loss / of dlogits (== batch on one GPU) */
int tcr_net_forward_train(const tcr_net* net, const float* params, float* stats, const float* feat,
                          const float* labels, int batch, int global_batch, float keep_prob,
                          uint64_t seed, int64_t sample_offset, float label_smoothing,
                          void* workspace, size_t workspace_bytes,
                          float* logits, float* probs, float* loss_out, void* stream);

/* Backward of the model loss wrt every trainable (tf.gradients inside
 * slim.learning.create_train_op, helper/trainer.py:205-211).  Must follow a forward_train on
 * the same workspace.  grads: arena-shaped, overwritten.  The L2 term is NOT added here
 * (it is folded into tcr_sgd_momentum_step / reported by tcr_l2_loss). */
int tcr_net_backward(const tcr_net* net, const float* params, const float* feat, int batch,
                     void* workspace, size_t workspace_bytes, float* grads, void* stream);

/* Cross-replica (sync) BN.  forward_train / backward can be run stage by stage: stage s ends right
 * after the per-channel partial sums of its BN layer ({sum y, sum y^2} forward, {sum dz, sum dz*xhat}
 * backward; 2*C float64 values, the per-workgroup partial rows added up in the fixed order the unstaged
 * path uses) have been written to a contiguous device buffer, which the host all-reduces (sum) across
 * replicas before calling stage s+1.  Statistics then span global_batch, exactly as in the single-device
 * reference; with one replica the staged run is bitwise the unstaged one.  Stages 0 .. tcr_net_num_stages()-1; the last stage
 * has no hand-off.  See DESIGN.md "Data parallel". */
int tcr_net_num_stages(const tcr_net* net, int backward);
int tcr_net_stage_sums(const tcr_net* net, int backward, int stage, void* workspace, int batch,
                       double** sums_dev, int64_t* n_doubles);
int tcr_net_forward_train_stage(const tcr_net* net, const float* params, float* stats, const float* feat,
                                const float* labels, int batch, int global_batch, float keep_prob,
                                uint64_t seed, int64_t sample_offset, float label_smoothing,
                                void* workspace, size_t workspace_bytes,
                                float* logits, float* probs, float* loss_out, int stage, void* stream);
int tcr_net_backward_stage(const tcr_net* net, const float* params, const float* feat, int batch, int global_batch,
                           void* workspace, size_t workspace_bytes, float* grads, int stage, void* stream);

/* The same hand-off by DEPENDENCY LEVEL (round 3): the units whose statistics become available together are handed over at once --
 * forward: conv0 | per block (down, conv_a) | conv_b; backward: per block, last first, (conv_b, down) | conv_a; then conv0 -- so a
 * training step needs 2 x (1 + 2 x blocks) all-reduces instead of 2 x (BN units): TCResNet8 14 instead of 20, TCResNet14 26 instead
 * of 32.  Levels 0 .. tcr_net_num_levels() - 1; after every level but the last the caller all-reduces (sum) the *n_doubles float64
 * values at *sums_dev (tcr_net_level_sums).  One replica: bitwise the unstaged run. */
int tcr_net_num_levels(const tcr_net* net, int backward);
int tcr_net_level_sums(const tcr_net* net, int backward, int level, void* workspace, int batch, double** sums_dev, int64_t* n_doubles);
int tcr_net_forward_train_level(const tcr_net* net, const float* params, float* stats, const float* feat, const float* labels, int batch,
                                int global_batch, float keep_prob, uint64_t seed, int64_t sample_offset, float label_smoothing,
                                void* workspace, size_t workspace_bytes, float* logits, float* probs, float* loss_out, int level, void* stream);
int tcr_net_backward_level(const tcr_net* net, const float* params, const float* feat, int batch, int global_batch,
                           void* workspace, size_t workspace_bytes, float* grads, int level, void* stream);

/* ------------------------------------------------------------------------------------------ */
/* Network: DS-CNN S / M / L (audio_nets/ds_cnn.py:19-118), the depthwise-separable baseline     */
/* ------------------------------------------------------------------------------------------ */
typedef struct tcr_dscnn_cfg {
    int32_t h_in;                   /* frames (49 for 40/20 ms) */
    int32_t w_in;                   /* MFCC coefficients (--num_mfccs 10) */
    int32_t num_classes;
    int32_t depth;                  /* 64 / 172 / 276 (ds_cnn.py:20,29,37) */
    int32_t n_separable;            /* 4 / 4 / 5 separable blocks */
    int32_t conv1_kh, conv1_kw;     /* 10 x 4 */
    int32_t conv1_sh, conv1_sw;     /* (2,2) S; (2,1) M, L */
    int32_t ds1_sh, ds1_sw;         /* stride of conv_ds_1: (1,1) S; (2,2) M, L */
    float bn_decay;                 /* 0.96 (ds_cnn.py:107) */
    float bn_eps;                   /* 0.001 */
} tcr_dscnn_cfg;

typedef struct tcr_dscnn tcr_dscnn;

int tcr_dscnn_create(const tcr_dscnn_cfg* cfg, tcr_dscnn** out);
void tcr_dscnn_destroy(tcr_dscnn* net);
int64_t tcr_dscnn_param_floats(const tcr_dscnn* net);   /* weights, biases, BN beta (no gamma: scale=False) */
int64_t tcr_dscnn_stat_floats(const tcr_dscnn* net);
int tcr_dscnn_num_tensors(const tcr_dscnn* net);
int tcr_dscnn_tensor_info(const tcr_dscnn* net, int index, tcr_tensor_info* out);   /* names: "DSCNN/conv_ds_1/pointwise_conv/weights", ... */
size_t tcr_dscnn_workspace_bytes(const tcr_dscnn* net, int batch);
/* Eval-mode forward: feat = front-end output [batch][w_in][tcr_padded_len(h_in)] (num_mfccs = w_in);
 * logits / probs [batch][num_classes].  DSCNN() + slim.softmax (ds_cnn.py:89-101, factory/audio_nets.py:147-156). */
int tcr_dscnn_forward_infer(const tcr_dscnn* net, const float* params, const float* stats, const float* feat,
                            int batch, void* workspace, size_t workspace_bytes, float* logits, float* probs, void* stream);

/* Train-mode forward of DSCNN() (is_training=True: batch statistics, moving averages updated with decay 0.96 --
 * DSCNN_arg_scope, ds_cnn.py:104-118) + softmax cross-entropy (factory/audio_nets.py:161-173); no dropout is applied
 * in the graph (ds_cnn.py:89-101).  Arguments as tcr_net_forward_train; the workspace keeps what backward needs. */
size_t tcr_dscnn_train_workspace_bytes(const tcr_dscnn* net, int batch);
int tcr_dscnn_forward_train(const tcr_dscnn* net, const float* params, float* stats, const float* feat, const float* labels,
                            int batch, int global_batch, float label_smoothing, void* workspace, size_t workspace_bytes,
                            float* logits, float* probs, float* loss_out, void* stream);
/* Gradient of the model loss wrt every trainable of the arena (tf.gradients inside slim.learning.create_train_op,
 * helper/trainer.py:205-211; the reference trains DS-CNN with Adam -> tcr_adam_step).  The conv / depthwise /
 * pointwise biases feed a train-mode BN, so their gradient is identically zero and is written as 0. */
int tcr_dscnn_backward(const tcr_dscnn* net, const float* params, const float* feat, int batch,
                       void* workspace, size_t workspace_bytes, float* grads, void* stream);
/* Cross-replica (sync) BN for DS-CNN, as tcr_net_*_stage: stage u of the forward ends with unit u's {sum y, sum y^2} (2*C float64) in
 * the hand-off buffer, stage k of the backward with {sum dz, sum dz*xhat} of unit (units-1-k); the host all-reduces them between stages.
 * tcr_dscnn_num_stages() stages each way; with one replica the staged run is bitwise the unstaged one. */
/* Where a training forward left the post-BN+ReLU activation of BN unit `unit` (0 = conv_1, then depthwise / pointwise of every
 * separable block) inside the caller's training workspace: [batch][channels][padded] floats at float offset *offset, the `positions`
 * values of a plane behind a TCR_HALO-float halo.  The reference's `endpoints` of ds_cnn.py:46-62,104-118; used by the parity tests
 * to take the float64 oracle's gradient on the kernels' side of ReLU inputs within round-off of zero. */
int tcr_dscnn_num_units(const tcr_dscnn* net);
int tcr_dscnn_unit_output(const tcr_dscnn* net, int unit, int batch, int64_t* offset, int* channels, int* positions, int* padded);
/* The default training path never writes that activation (its consumers apply BN + ReLU to the unit's raw conv output as they read it,
 * TCR_TUNE_DS_TRAIN): this call computes it from the raw output and the batch statistics the last training forward left in the workspace,
 * into the slot tcr_dscnn_unit_output() names.  Replaces nothing in the reference (test / inspection hook for `endpoints`). */
int tcr_dscnn_materialize_unit(const tcr_dscnn* net, int unit, int batch, void* workspace, size_t workspace_bytes, void* stream);
int tcr_dscnn_num_stages(const tcr_dscnn* net);
int tcr_dscnn_stage_sums(const tcr_dscnn* net, int backward, int stage, void* workspace, int batch, double** sums_dev, int64_t* n_doubles);
int tcr_dscnn_forward_train_stage(const tcr_dscnn* net, const float* params, float* stats, const float* feat, const float* labels,
                                  int batch, int global_batch, float label_smoothing, void* workspace, size_t workspace_bytes,
                                  float* logits, float* probs, float* loss_out, int stage, void* stream);
int tcr_dscnn_backward_stage(const tcr_dscnn* net, const float* params, const float* feat, int batch, int global_batch,
                             void* workspace, size_t workspace_bytes, float* grads, int stage, void* stream);

/* ------------------------------------------------------------------------------------------ */
/* Generic 2-D layer graph: the other model families behind the factory        */
/*   ResNet2D8 / ResNet2D8Pool (audio_nets/tc_resnet.py:14-15,23-24,73-99),                       */
/*   Res8 / Res8Narrow / Res15 / Res15Narrow (audio_nets/res.py:6-123),                           */
/*   KWSModel architectures (audio_nets/kws.py:15-63).                                            */
/* ------------------------------------------------------------------------------------------ */
/* The host describes the topology node by node, as the reference's Python builds its TF graph; every builder returns the
 * node id (>= 0) or a negative tcr_status.  Input id -1 is the network input [batch][c][h*w + 2*TCR_HALO] (planar, the
 * plane at offset TCR_HALO).  Variables live in one trainable arena (all variables whose TF name lacks "BatchNorm" first:
 * the L2-regularised set of factory/audio_nets.py:175-180) and one moving-statistics arena, under the names given here. */
typedef struct tcr_g2d tcr_g2d;
int tcr_g2d_create(const char* scope, int h, int w, int c, tcr_g2d** out);
void tcr_g2d_destroy(tcr_g2d* g);
/* slim.conv2d / tf.nn.conv2d (+ bias, + ReLU): kernel kh x kw, stride, dilation `rate` (audio_nets/res.py:11-16), SAME
 * (valid_padding = 0) or VALID padding; weights HWIO under `weights_name`, bias under `biases_name` (NULL / "": none).
 * A fully connected layer over a flattened [h][w][c] activation is the VALID conv with kh = h, kw = w. */
int tcr_g2d_conv(tcr_g2d* g, int in, int kh, int kw, int cout, int sh, int sw, int dh, int dw, int valid_padding, int relu,
                 const char* weights_name, const char* biases_name);
/* slim.batch_norm(fused): optional beta (center) / gamma (scale), optional ReLU; variables `<prefix>/gamma|beta|moving_*`. */
int tcr_g2d_batch_norm(tcr_g2d* g, int in, int center, int scale, int relu, float decay, float eps, const char* prefix);
/* slim.avg_pool2d / tf.nn.max_pool; kh <= 0: the window is the whole plane (global pool). */
int tcr_g2d_pool(tcr_g2d* g, int in, int is_max, int kh, int kw, int sh, int sw, int valid_padding);
int tcr_g2d_add(tcr_g2d* g, int a, int b, int relu);                 /* net += layer_in [; relu] */
int tcr_g2d_dropout(tcr_g2d* g, int in, float keep_prob);            /* tf.nn.dropout / slim.dropout; identity in eval mode */
/* SVDF layer of KWSModel --architecture low_latency_svdf (audio_nets/kws.py:490-680, training graph): after the frequency filters (a
 * 1 x F VALID conv) `time_filter` applies one filter of the plane's length per channel (tf.matmul with weights_time [filters, T], :604-612),
 * `group_sum` adds the `rank` filters of a unit, the bias and the ReLU (:613-628). */
int tcr_g2d_time_filter(tcr_g2d* g, int in, const char* weights_name);
int tcr_g2d_group_sum(tcr_g2d* g, int in, int group, int relu, const char* biases_name);
int tcr_g2d_node_shape(const tcr_g2d* g, int node, int* c, int* h, int* w);
/* A node's activation inside the workspace of a forward call at (batch, train): [batch][C][plane_floats], the H*W values of a
 * plane start `halo` floats in.  What the reference exposes as `endpoints` (audio_nets/res.py:66, tc_resnet.py:95). */
int tcr_g2d_node_output(const tcr_g2d* g, int node, int batch, int train, int64_t* offset_floats, int64_t* plane_floats, int* halo);
int tcr_g2d_finalize(tcr_g2d* g, int logits_node);                   /* logits node: [num_classes] x 1 x 1 */
int64_t tcr_g2d_param_floats(const tcr_g2d* g);
int64_t tcr_g2d_decay_floats(const tcr_g2d* g);
int64_t tcr_g2d_stat_floats(const tcr_g2d* g);
int tcr_g2d_num_tensors(const tcr_g2d* g);
int tcr_g2d_num_classes(const tcr_g2d* g);
int tcr_g2d_tensor_info(const tcr_g2d* g, int index, tcr_tensor_info* out);
size_t tcr_g2d_workspace_bytes(const tcr_g2d* g, int batch, int train);
/* front-end output [batch][f][tcr_padded_len(t)] -> the [t x f] single-channel plane these networks read ([N, T, F, 1]) */
int tcr_g2d_input_from_features(const float* feat, int batch, int t, int f, float* plane, void* stream);
/* eval forward + softmax; train forward (batch statistics, dropout keyed by (seed, node, sample), mean cross-entropy:
 * arguments as tcr_net_forward_train); backward of the model loss wrt every trainable (same seed / sample_offset). */
int tcr_g2d_forward_infer(const tcr_g2d* g, const float* params, const float* stats, const float* x, int batch,
                          void* workspace, size_t workspace_bytes, float* logits, float* probs, void* stream);
int tcr_g2d_forward_train(const tcr_g2d* g, const float* params, float* stats, const float* x, const float* labels, int batch,
                          int global_batch, uint64_t seed, int64_t sample_offset, float label_smoothing, void* workspace,
                          size_t workspace_bytes, float* logits, float* probs, float* loss_out, void* stream);
int tcr_g2d_backward(const tcr_g2d* g, const float* params, const float* x, int batch, uint64_t seed, int64_t sample_offset,
                     void* workspace, size_t workspace_bytes, float* grads, void* stream);
/* Cross-replica BN for the graph engine (round 3): the forward / backward stop behind every BN node's statistics (graph order / reverse
 * graph order), the caller all-reduces the 2 x C float64 sums (tcr_g2d_stage_sums) and runs the next stage; stages
 * 0 .. tcr_g2d_num_stages() - 1 (= BN nodes + 1).  One replica: bitwise the unstaged run. */
int tcr_g2d_num_stages(const tcr_g2d* g);
int tcr_g2d_stage_sums(const tcr_g2d* g, int backward, int stage, void* workspace, int batch, double** sums_dev, int64_t* n_doubles);
int tcr_g2d_forward_train_stage(const tcr_g2d* g, const float* params, float* stats, const float* x, const float* labels, int batch,
                                int global_batch, uint64_t seed, int64_t sample_offset, float label_smoothing, void* workspace,
                                size_t workspace_bytes, float* logits, float* probs, float* loss_out, int stage, void* stream);
int tcr_g2d_backward_stage(const tcr_g2d* g, const float* params, const float* x, int batch, int global_batch, uint64_t seed,
                           int64_t sample_offset, void* workspace, size_t workspace_bytes, float* grads, int stage, void* stream);

/* ------------------------------------------------------------------------------------------ */
/* Optimiser (helper/trainer.py:171-197) and L2 (factory/audio_nets.py:175-182)                */
/* ------------------------------------------------------------------------------------------ */
/* tf.train.MomentumOptimizer, use_nesterov=False:  g' = g*grad_scale + wd*w (first n_decay floats)
 * a <- mu*a + g' ; w <- w - lr*a. */
int tcr_sgd_momentum_step(float* params, const float* grads, float* momentum, int64_t n, int64_t n_decay,
                          float lr, float mu, float weight_decay, float grad_scale, void* stream);
/* tf.train.AdamOptimizer (beta1 .9, beta2 .999, eps 1e-8 defaults); t = 1-based step. */
int tcr_adam_step(float* params, const float* grads, float* m, float* v, int64_t n, int64_t n_decay,
                  float lr, float beta1, float beta2, float eps, int64_t t, float weight_decay,
                  float grad_scale, void* stream);
/* tf.train.RMSPropOptimizer (decay .9, momentum 0, eps 1e-10 defaults; helper/trainer.py:186-188):
 * ms <- decay*ms + (1-decay)*g'^2 ; mom <- momentum*mom + lr*g'/sqrt(ms + eps) ; w <- w - mom.  The `ms` slot starts at one. */
int tcr_rmsprop_step(float* params, const float* grads, float* ms, float* mom, int64_t n, int64_t n_decay,
                     float lr, float decay, float momentum, float eps, float weight_decay, float grad_scale, void* stream);
/* tf.train.ExponentialMovingAverage(decay).apply(variables_to_train) (helper/trainer.py:213-217):
 * shadow <- shadow - (1 - decay) * (shadow - params), over the trainable arena. */
int tcr_ema_step(float* shadow, const float* params, int64_t n, float decay, void* stream);
/* out[0] = weight_decay * sum_{i<n_decay} 0.5*w_i^2 (device float). */
int tcr_l2_loss(const float* params, int64_t n_decay, float weight_decay, float* out, void* stream);
/* Batch SUM of the softmax cross-entropy of logits rows [batch][num_classes] against one-hot (optionally smoothed) labels: the model
 * loss of an evaluation build (tf.losses.softmax_cross_entropy, factory/audio_nets.py:161-173) times the batch. */
int tcr_xent_loss_sum(const float* logits, const float* labels, int batch, int num_classes, float label_smoothing,
                      float* loss_utt, float* loss_sum, void* stream);

/* ------------------------------------------------------------------------------------------ */
/* Input stage: PCM decode + crop/pad + time shift + background mix            */
/* ------------------------------------------------------------------------------------------ */
/* Batched device form of contrib_audio.decode_wav(desired_samples) + _shift_audio + _mix_background
 * (datasets/augmentation_factory.py:30-211; mapped per element by AudioDataWrapper._parse_function,
 * datasets/audio_data_wrapper.py:37-58).  The random draws of the reference's graph are inputs:
 *   pcm        int16 pool holding every clip (mono, already at sample_rate);
 *   clip_off   [batch] first sample of each utterance's clip in the pool;
 *   clip_len   [batch] decoded samples of the clip (0 = the empty filename of a "silent" sample); longer clips are
 *              cropped, shorter ones zero-padded to desired_samples (decode_wav).  NULL: every clip has desired_samples;
 *   shift      [batch] time_shift_amount in samples, + delays the audio (zero fill, _shift_audio :104-141); NULL: 0;
 *   background int16 pool of the background-noise recordings; bg_off [batch] first sample of the random crop;
 *   bg_vol     [batch] background_volume (0 = not mixed: not read); NULL: no mixing at all;
 *   out        [batch][desired_samples] float32 = clip(background / 32768 * bg_vol + foreground, -1, 1)  (:92-95).
 * One IEEE multiply and one IEEE add per sample, like tf.multiply / tf.add: results are bit-exact. */
int tcr_augment_fwd(const int16_t* pcm, const int64_t* clip_off, const int32_t* clip_len, const int32_t* shift,
                    const int16_t* background, const int64_t* bg_off, const float* bg_vol, int batch,
                    int desired_samples, float* out, void* stream);

/* ------------------------------------------------------------------------------------------ */
/* Instrumentation                                                                             */
/* ------------------------------------------------------------------------------------------ */
/* Name of the n-th kernel family in this library (NULL past the end); used by bench.py to match
 * rocprofv3 kernel-trace rows. */
const char* tcr_kernel_name(int index);

/* Process-wide kernel-selection knobs for A/B measurements (defaults = 0 = the tuned choice). */
enum { TCR_TUNE_CONV_PATH = 0,   /* 0 auto: implicit-GEMM MFMA conv where the shape fits, 1: scalar-fed VALU conv, 2: as 0 */
       TCR_TUNE_FRONTEND = 1,    /* 0 / 5: packed-FP32 kernel (default); 1..4: scalar-FP32 kernel, variant (v-1): bit0 wave-local phase ordering, bit1 sample prefetch */
       TCR_TUNE_CONV_B = 2,      /* MFMA conv activations: 0 straight from global/L1 (default), 1 via an LDS image; 3: wide 1x1 convs on the register-fed kernel instead of the LDS-tiled one, DS-CNN conv_1 not fused with the first depthwise layer */
       TCR_TUNE_NET_FUSED = 3,   /* eval forward: 0 one fused LDS-resident kernel for the whole net (default; layers of the flagship shapes run compile-time-specialised), 1 per-layer kernels, 2 fused with the features copied to LDS, 3 fused, generic layer walk only, 4 the round-2 static-shape layer, 5 the static kernel with four 16-position tiles per job in block 0's layers instead of two (round 5 experiment: bitwise, no faster), 7 fused without the bank-aligned utterance strides (A/B arm); the static kernels' nine-tap layers (round 6): 0 work dealt in 16-position units, as even over the waves as units allow, + a whole tap of weight lookahead in the layers of <= 32 input channels (TCResNet14-1.5: <= 48) (default), 8 jobs of two tiles dealt round-robin (rounds 3-5), 9 units without the lookahead (all bitwise) */
       TCR_TUNE_FUSED_GROUP = 4, /* utterances per workgroup group of the fused kernel (0: largest that fits 64 KB of LDS) */
       TCR_TUNE_FUSED_WAVES = 5, /* fused kernel: waves per workgroup (4, 8, 16) + 100 * weight-ring depth (4, 8, 16); 0: default */
       TCR_TUNE_CONV_KSPLIT = 6, /* train-mode conv / data-gradient: waves sharing one 32-position group's reduction (0 auto, 1, 2, 4) */
       TCR_TUNE_WGRAD_STREAM = 7,/* backward: 0 weight-gradient kernels on the library's internal streams (one set per device and process; default), 1 everything on the caller's stream, 2: as 0 with the TC-ResNet shortcut units (BN backward, data and filter gradient) on the second internal stream instead of behind the other units' filter gradients (measured: -1 % at 49 frames, +7 % for TCResNet8 at 98); 3: lazy backward with one stream fork per block instead of one per BN unit (measured slower: +3 %) */
       TCR_TUNE_TRAIN_FWD = 8,   /* train-mode forward: 0 group-resident phases (train_fused.hip; BN affine / ReLU / residual applied while the next conv stages its input, statistics from the conv epilogue), 1 per-layer kernels (conv -> statistics -> finalize -> normalise), 2: as 0 with the head walking the block output's rows for its pooling instead of starting from the sums over time the closing phase leaves (round 6; bitwise the same) */
       TCR_TUNE_TRAIN_BWD = 9,   /* TC-ResNet backward: 0 "lazy" BN backward (bwd_lazy.hip: dy never written -- the data-gradient kernel applies BN backward while it stages a group of utterances into LDS, runs every stride phase and the block's shortcut conv from that image and leaves the next unit's sums from its epilogue; the filter-gradient kernels compute dy where they load it; default for nets of <= 48 channels, where it measured faster; 3: for every net it covers), 1 the group-resident phases of round 2 (train_fused_bwd.hip), 2 the per-layer chain (reduce -> finalize + bn_bwd_apply -> data gradient per phase; the default until round 3) */
       TCR_TUNE_PHASE_CFG = 10,  /* training phases: waves per workgroup * 100 + utterances per group (0: default) */
       TCR_TUNE_BWD_BN_FUSED = 11, /* BN backward: 0 finalize folded into the apply pass (one launch, ~512 workgroups; round 6: for every layer width -- rounds 3-5: <= 48 channels, 1024 workgroups), 1 finalize + apply kernels, >= 2: folded, that many workgroups aimed at */
       TCR_TUNE_BWD_MASK = 12,   /* BN backward: 0 a unit's own ReLU mask recomputed from its raw conv output ([fmaf(y, scale, shift) > 0], bitwise the activation's; default), 1 read back from the stored activation, 2: as 0 with the scalar (one element per thread) elementwise BN kernels instead of the 16-byte ones (bitwise the same), 3: also the scalar per-channel reduction kernel (another summation order), 4: the 16-byte reduction kernel also where its grid would be small (tests), 5: as 0 with the lazy backward's last block reduced by two launches (conv_b's unit, then the shortcut's reading the masked gradient back) instead of one two-unit pass (round 6; bitwise the same rows) */
       TCR_TUNE_FE_GRID = 13,    /* front-end: cap on the number of persistent workgroups (0: two per CU). 256 = one per CU, which leaves half of every CU's LDS and registers to a co-resident network kernel on another stream */
       TCR_TUNE_FUSED_GRID = 14, /* fused eval network: cap on the number of persistent workgroups (0: as many as the LDS allows per CU) */
       TCR_TUNE_DS_TRAIN = 15,   /* DS-CNN training: 0 normalised activations never materialised where every consumer has the form (172 / 276-channel nets): consumers apply BN + ReLU to the raw conv outputs, batch statistics and backward sums come from conv / data-gradient epilogues (default); 1 the materialising path (statistics reduce -> finalize -> normalise, backward reduce); 2: as 0, but every unit's BN backward by a bn_bwd_apply pass (default 0: conv_1's filter gradient computes dy where it reads it); 3: as 0, the depthwise units' kernels too; 4: as 0, and the pointwise units' data-gradient kernel applies the BN backward while staging and writes dy for the filter gradient instead of a bn_bwd_apply pass (measured slower) */
       TCR_TUNE_WGRAD_TILES = 16, /* 9-tap filter gradients (16-byte-load kernel): output-channel tiles per launch (0: default 2 since round 6, 3 before; a layer of more tiles is split into launches that share one slab) */
       TCR_TUNE_DOWN_DGRAD = 17,  /* TC-ResNet backward, a block's 1x1 shortcut conv: 0 its data gradient runs early on the side stream and writes the block-input gradient first, conv_a's adds onto it (default for nets of <= 48 channels and, since round 6, for wider nets from 64 frames up, where it measured faster; 2: for every net); 1 conv_a's first, the shortcut's added behind it on the main stream (bitwise the same sums: one addition, commuted) */
       TCR_TUNE_BWD_LAZY_CFG = 18, /* lazy backward geometry: utterances per group + 100 * waves per job (0: cost model) + 10000 * (out channels * 10 + layers) to address one kernel of the net */
       TCR_TUNE_PHASE_STATIC = 19, /* training forward phases of TCResNet8-1.0 / TCResNet14-1.5 at 49 / 98 frames: 0 compile-time-shaped kernels, utterance stride in LDS padded to the bank pattern (default); bit 0: generic layer walk; bit 1: unpadded stride; bit 2: the phases' staging one element at a time with its coefficients gathered from global memory (rounds 2-5) instead of float4 accesses + an LDS coefficient table (round 6) (A/B arms, all bitwise) */
       TCR_TUNE_WGRAD_WAVES = 20, /* on-the-fly 9-tap filter gradients: waves per workgroup (0: policy; 4, 8, 12, 16) */
       TCR_TUNE_WGRAD_LDS = 21,  /* first conv's filter gradient: 0 the LDS-staged nine-wave kernel (default), 1 the 16-byte-load kernel; per-layer chain, round 6 A/B arms (bitwise the default): 2 the first conv's dy written by an apply pass instead of built where the filter gradient loads it, 3 every layer's split-K slabs summed in one pass at the step's end instead of the first half of the units early */
       TCR_TUNE_LAZY_STAGE = 22, /* lazy backward: 0 the group's rows staged with 16-byte loads (default), 1 a dword gather per interior element (bitwise the same) */
       TCR_TUNE_FE_KERNEL = 23,  /* packed-FP32 front-end: 0 the three-waves-per-SIMD kernel (frontend_pk3.hip: wave-local LDS regions, <= 168 registers; default), 1 the two-waves kernel of rounds 2-4 (frontend_pk.hip; also what filterbanks with more work items than the unrolled trips fall back to). Bitwise the same features. */
       TCR_TUNE_FE_STAGGER = 24, /* three-waves front-end: one-off start-up delay of (workgroup generation * 4 + wave) * value * 64 cycles that de-phases the twelve waves of a CU (0: none) */
       TCR_TUNE_PW_WGRAD = 25,   /* wide pointwise (DS-CNN 172 / 276 channels) filter gradient: 0 the register-staged kernel (two 4-wave workgroups per CU; on 13 x 5 maps its unrolled form with fixed staging roles, round 5; default), 1 the DMA-staged kernel (global_load_lds into two LDS buffers, one 12-wave workgroup per CU, three split-K wave groups; measured 2 % slower), 2 the register-staged kernel's run-time-shape form on every map (rounds 4-5; bitwise the default) */
       TCR_TUNE_DEPLOY_F32 = 26, /* deploy-path MFCC (method 2): 0 the float64 kernel (one workgroup per frame; TF's ops compute in double; default), 1 the float32 throughput kernels with the op's filterbank / log floor (rounds 3-4: up to 0.5 off on noise-free tones, where the empty bands are pure round-off) */
       TCR_TUNE_NET_SMALL = 27,  /* eval network, TCResNet8-1.0 at 49 frames, batches of <= 64 utterances: 0 the small-batch kernel (one utterance per 8-wave workgroup, each phase's weights DMA-copied into LDS one phase ahead; default), 1 the throughput kernel at one utterance per group (rounds 2-4).  Bitwise the same outputs. */
       TCR_TUNE_PW_POS = 28,     /* wide pointwise convs (DS-CNN-L, 276 channels; forward, data gradient): 0 the nine-tile kernel built for <= 128 registers = four waves per SIMD, its weight chunks copied global -> LDS by the DMA path and its LDS fragment reads one step ahead of the MFMAs (default since round 5), 1 the unconstrained build of rounds 3-4 (92 VGPRs + 72 AGPRs, three waves per SIMD, register-staged weights), 2 the <= 128-register build with register-staged weights.  Bitwise the same results. */
       TCR_TUNE_BN_APPLY = 29,   /* BN-backward apply passes: 0 four float4 per thread and operand, per-channel coefficients staged in LDS (large tensors, DS-CNN: default since round 5; the finalize-folded pass of the TC-ResNet chain, bn_bwd_apply_fused_kernel: round 6), 1 the one-float4-per-thread loops of rounds 2-5.  Bitwise the same dy. */
       TCR_TUNE_DW_DGRAD = 30,   /* DS-CNN depthwise data gradient, stride-1 units on 13 x 5 maps: 0 the row kernel (dz / raw / dx blocks of 16 planes as contiguous float4 through LDS, one lane per map row; default since round 5), 1 the zero-padded-image kernel of rounds 2-4.  Bitwise the same dx and backward sums. */
       TCR_TUNE_DW_WGRAD = 31,   /* DS-CNN depthwise filter gradient, stride-1 units on 13 x 5 maps: 0 the row kernel (a wave owns four channels, their x / dz planes as contiguous float4 through wave-private LDS; default since round 5), 1 the gather kernel of rounds 2-4 (another summation order: equal to rounding). */
       TCR_TUNE_DW_FWD = 32,     /* DS-CNN depthwise conv (eval and training forward), stride-1 layers on 13 x 5 maps: 0 the row kernels (x / y blocks of 16 planes as contiguous float4 through LDS; also the global pooling's block-copy kernel and the row-per-lane stencil of the fused conv_1 + depthwise eval kernel; default since round 5), 1 the zero-padded-image kernels of rounds 2-4.  Bitwise the same outputs and statistics. */
       TCR_TUNE_WGRAD_PIPE = 33, /* 16-byte-load filter gradients (conv_wgrad_mfma4_kernel): 0 software-pipelined -- the operands of a wave's NEXT trip (the next 16 / 8 positions, or the next utterance's first) are requested before the current trip's MFMAs, two register sets alternating (round 6; bitwise the old kernel: same trips, same order; default), 1 every trip loads, waits, multiplies (rounds 3-5) */
       TCR_TUNE_COUNT = 34 };
int tcr_tune(int knob, int value);

/* The library's internal streams (hipStream_t), one set per device and process.  HIP multiplexes streams onto a few hardware queues
 * and streams that share a queue serialise; the set is chosen on first use -- candidates are probed -- so that streams 0, 1, 2 and
 * `caller_stream` (the stream the host launches on; it is synchronised a few times by that first call) run concurrently, and stream 3
 * runs concurrently with `caller_stream` and stream 2.  0 / 1: the backward's filter gradients, used internally.  2: for the host's
 * input stage (the next batch's front-end next to a training step: the reference's tf.data prefetch, datasets/data_wrapper_base.py:70-76).
 * 3: for the network of a two-stream inference pipeline whose front-end runs on 2.  NULL on failure or idx outside 0..3. */
void* tcr_internal_stream(int idx, void* caller_stream);

#ifdef __cplusplus
}
#endif
#endif /* TCRESNET_HIP_H */
