/* pbsed.h - C-ABI of libpbsed_mi355.so: MI355X (gfx950) kernels for the pb_sed FBCRNN / BiCRNN hot path.
 *
 * The reference (fgnt/pb_sed) is pure Python and reaches its kernels through torch ops; it has no FFI of
 * its own.  Each entry point below therefore replaces a *torch-level op site* of the reference and cites
 * it (paths relative to the reference repo).  Conventions:
 *   - all pointers are DEVICE pointers owned by the caller unless marked "host"; the library never frees or
 *     retains them; `stream` is a hipStream_t (NULL = default stream); every call is asynchronous.
 *   - activations use the reference layouts: 2-D [B, C, F, T], 1-D [B, C, T] (passed as F = 1);
 *     GRU scan buffers are time-major [T, B, *].
 *   - nothing is process-wide: kernel attributes, CU counts and occupancy figures are cached per device ordinal, so one
 *     process may drive several GPUs.  The only memory the library owns is the partial-sum scratch of the weight-gradient
 *     kernels (convolution slots, GRU slots): per DEVICE by default - then calls for one device must come from one stream
 *     at a time - or the caller's own per (device, stream) after pbsed_set_scratch().  The persistent-scan workspaces and
 *     every other buffer are arguments.
 *   - every function returns 0 on success or a negative PBSED_E_* code; pbsed_last_error() returns the
 *     thread-local message.  No C++ exception crosses the boundary.
 */
#ifndef PBSED_H
#define PBSED_H
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PBSED_OK 0
#define PBSED_E_ARG (-1)
#define PBSED_E_HIP (-2)
#define PBSED_E_UNSUPPORTED (-3)
/* statistics accumulators (`stats`, `sums`) are [PBSED_STAT_SLOTS][C][2] doubles, zeroed by the caller;
 * kernels spread their atomics over the slots, the *_finalize entry points sum them. */
#define PBSED_STAT_SLOTS 32

const char* pbsed_last_error(void);
int pbsed_version(void);
/* Caller-owned scratch for (current device, stream): pbsed_conv_bwd_weight* and pbsed_gru_wgrad* take their partial-sum slots
 * from it instead of the library's per-device buffer (several streams of one device may then run them concurrently).
 * scratch = NULL removes the registration; pbsed_scratch_bytes() covers every launch of the reference networks.  The
 * library fills the first 2 MB of a registered buffer with zeros once and KEEPS them zero between its launches (the slotted
 * convolution gradients add into them, their reduction pass clears them again): hands off while it is registered. */
size_t pbsed_scratch_bytes(void);
int pbsed_set_scratch(void* scratch /*device*/, size_t bytes, void* stream);
/* CU budget for grid sizing: the weight-gradient entry points (pbsed_conv_bwd_weight*, pbsed_gru_wgrad*) cut their persistent
 * grids for `cus` compute units instead of the whole device until the budget is changed again (0 = whole device; process-wide,
 * set and reset around the launches it is meant for).  For launches that are enqueued BESIDE a persistent GRU scan on another
 * stream (engine.SIDE_WGRAD): the scan holds up to 7/8 of the CUs for its whole duration, and a grid cut for the full device
 * would leave most of its workgroups waiting for the scan to end.  Returns the previous budget.  Results do not depend on it
 * beyond the order of the partial sums.  No counterpart in the reference (single stream, library kernels). */
int pbsed_set_launch_cus(int cus);

/* ---- fused front-end.  Replaces the CPU STFT (pb_sed/data_preparation/provider.py:315-323, called at
 * pb_sed/data_preparation/transform.py:53) + NormalizedLogMelExtractor (pb_sed/models/weak_label/crnn.py:86-90).
 * wav [B, n_samples] -> out [B, 1, F, T];  window[960], twiddle[1024][2], sparse mel filters
 * (mel_start/mel_len/mel_off [F], mel_w flat with mel_w_count <= 1152 entries), mean/inv_std [F]; seq_len_frames [B] or NULL.
 * F <= 512, n_samples < 2^29 (the kernel addresses a clip with 32-bit byte offsets); PBSED_E_UNSUPPORTED beyond. */
int pbsed_logmel_fwd(const float* wav, int B, int n_samples, int T, const int* seq_len_frames,
                     const float* window, const float* twiddle, const int* mel_start, const int* mel_len,
                     const int* mel_off, const float* mel_w, int mel_w_count, int F, const float* mean, const float* inv_std,
                     float eps, float clampv, float* out, double* stats, int pad_front, const float* mel_pts,
                     void* stream);
/* The same front-end with an explicit first-sample index per frame, frame_pos [B][T] int32 (windows reaching outside
 * [0, n_samples) read zeros): the time-warped STFT of the training pipeline (TimeWarpedSTFT at
 * pb_sed/data_preparation/transform.py:36-45 with the samplers of provider.py:329-338) computed on the GPU - the host
 * draws the warp and hands over where each frame starts, see pb_sed_amd/data.py::TimeWarp. */
int pbsed_logmel_fwd_frames(const float* wav, int B, int n_samples, int T, const int* seq_len_frames, const int* frame_pos,
                            const float* window, const float* twiddle, const int* mel_start, const int* mel_len,
                            const int* mel_off, const float* mel_w, int mel_w_count, int F, const float* mean,
                            const float* inv_std, float eps, float clampv, float* out, double* stats, const float* mel_pts,
                            void* stream);
/* `mel_pts` (both front-end entry points): NULL = the static sparse filterbank; else [B][F+2] fractional STFT-bin
 * positions of each clip's (warped) triangular filters - filter m spans mel_pts[b][m] .. [m+2] with its peak at [m+1],
 * unit sum: the per-example MelWarping of the training config (pb_sed/experiments/weak_label_crnn/training.py:195-208).
 * `pad_front`: zero samples assumed before wav[0]; frame t covers samples [320 t - pad_front, +960).  320 is the
 * reference's 'half' fading (provider.py:315-323); 0 for a slice cut out of the middle of a long clip.
 * `stats` (both front-end entry points): NULL, or [PBSED_STAT_SLOTS][F][2] zeroed doubles that receive the per-mel sum
 * and sum of squares of the values written for frames < seq_len - the training-mode statistics pass of the feature
 * normalisation (call with mean = 0, inv_std = 1, clampv = inf, then pbsed_feature_norm_update, then
 * pbsed_augment_logmel(mean, inv_std, clampv) to normalise in place).
 * The reference's own input contract: inputs['stft'] [B,1,T,bins,2] fp32 complex STFT computed by the CPU data loader
 * (pb_sed/models/weak_label/crnn.py:31,79-83 pops it; feature extractor call :86-90) -> |X|^2 -> sparse mel -> log ->
 * (x - mean) * inv_std -> clamp -> frames >= seq_len zeroed -> out [B, 1, F, T].  Same tables as pbsed_logmel_fwd. */
int pbsed_logmel_from_stft(const float* stft, int B, int T, int bins, const int* seq_len_frames,
                           const int* mel_start, const int* mel_len, const int* mel_off, const float* mel_w,
                           int mel_w_count, int F, const float* mean, const float* inv_std, float eps, float clampv,
                           float* out, double* stats, const float* mel_pts, void* stream);
/* Cumulative statistics of NormalizedLogMelExtractor's Normalization(statistics_axis='bt', momentum=None) (config
 * pb_sed/experiments/weak_label_crnn/training.py:190-217): running_mean / running_power [F] over all `num_tracked`
 * valid (clip, frame) positions so far are advanced by this batch's `count` positions with sums `stats`; mean and
 * inv_std = 1/sqrt(power - mean^2 + eps) [F] are what the normalising pass applies. */
int pbsed_feature_norm_update(const double* stats, double count, float* running_mean, float* running_power,
                              double* num_tracked, float eps, float* mean, float* inv_std, int F, void* stream);
/* Training-only feature augmentation (NormalizedLogMelExtractor config, pb_sed/experiments/weak_label_crnn/
 * training.py:209-216): x [B,F,T] in place; if mean != NULL first x = clamp((x - mean[f]) * inv_std[f], +-clampv) (the
 * normalisation of a statistics-tracking pass); x += noise_scale[b] * noise (noise may be NULL), then the time mask
 * [masks[b][0], masks[b][1]) and the frequency mask [masks[b][2], masks[b][3]) are zeroed (masks may be NULL), then
 * frames >= seq_len[b]. */
int pbsed_augment_logmel(float* x, const float* noise, const float* noise_scale, const int* masks /*[B][4]*/,
                         const int* seq_len, const float* mean, const float* inv_std, float clampv,
                         int B, int F, int T, void* stream);

/* ---- data front-end.  Scale + superposition mixing of resident waveforms
 * (pb_sed/data_preparation/mix.py:67-155, provider.py:195-215): out [B, n_out] = per clip the sum, in list order, of its
 * components comps[first[b] .. first[b+1]) - slices pool[src_off .. +length) placed at `start`, times `gain` (fp32), times
 * the raised-cosine fade of `fade_len` samples at the ends flagged fade_in / fade_out (float64, as numpy does).
 * Target encoding (pb_sed/data_preparation/transform.py:56-124): events of clip b = events[ev_first[b] .. ev_first[b+1])
 * (class, start frame, stop frame, type 0 weak / 1 boundaries / 2 strong) -> weak [B,K], boundary / strong [B,K,T]
 * (either may be NULL) with the reference's 0.5 conventions for unlabeled clips and weakly present classes. */
typedef struct { long long src_off; int length, start; float gain; int fade_in, fade_out; int pad_; } pbsed_mix_comp;
typedef struct { int cls, start, stop, type; } pbsed_target_event;
int pbsed_mix_clips(const float* pool, const pbsed_mix_comp* comps /*device*/, const int* first /*device [B+1]*/, float* out,
                    int B, int n_out, int fade_len, void* stream);
int pbsed_encode_targets(const pbsed_target_event* events /*device*/, const int* ev_first /*device [B+1]*/,
                         const int* unlabeled /*device [B]*/, const int* seq_len /*device [B]*/, float* weak, float* boundary,
                         float* strong, int B, int K, int T, void* stream);

/* ---- convolutions (CNN2d 3x3 / CNN1d k=1,3 / GRU input projections / heads): the `self.cnn(...)`,
 * `self.rnn_*` op sites pb_sed/models/weak_label/crnn.py:93,61-67; layer list
 * pb_sed/experiments/weak_label_crnn/training.py:159-169,218-260. */
void pbsed_conv_pack_dims(int KH, int KW, int Cin, int Cout, int dgrad, int* in_padded /*host*/,
                          int* out_padded /*host*/);
/* w [Cout,Cin,KH,KW] -> packed [KH*KW][in_padded][out_padded]; dgrad=1 packs the flipped/transposed form. */
int pbsed_pack_conv_weights(const float* w, float* w_packed, int Cout, int Cin, int KH, int KW, int dgrad,
                            void* stream);
/* y = conv(pad(mask(relu(x*scale+shift)))) + bias, optional (2,1) max-pool (pool_idx = argmax row) and
 * per-channel (or per (channel,f)) masked sum / sum-of-squares of y accumulated into stats [C][2] (double). */
int pbsed_conv_fwd(const float* x, const float* w_packed, const float* bias, const float* scale,
                   const float* shift, int relu, const int* seq_len, float* y, unsigned char* pool_idx,
                   double* stats, int stats_per_cf, int B, int Cin, int Cout, int F, int T, int KH, int KW,
                   int pool, void* stream);
/* Residual connections ('deep' net_config, pb_sed/experiments/weak_label_crnn/training.py:170-183; padertorch's skip
 * path restated, parity unpinned).  pbsed_conv_fwd_res: pbsed_conv_fwd whose (biased, pooled) output gets `residual`
 * [B,Cout,Fo,T] added before the store and before the statistics.  A skip that crosses a (2,1) pool is max-pooled with
 * pbsed_pool21_fwd (x as n_out row pairs of length T -> y, idx) and its gradient routed back with pbsed_pool21_bwd_add
 * (dx[argmax row] += g); pbsed_add_inplace: a += b. */
int pbsed_conv_fwd_res(const float* x, const float* w_packed, const float* bias, const float* scale,
                       const float* shift, int relu, const int* seq_len, float* y, unsigned char* pool_idx,
                       double* stats, int stats_per_cf, int B, int Cin, int Cout, int F, int T, int KH, int KW,
                       int pool, const float* residual, void* stream);
int pbsed_pool21_fwd(const float* x, float* y, unsigned char* idx, size_t n_out, int T, void* stream);
int pbsed_pool21_bwd_add(const float* g, const unsigned char* idx, float* dx, size_t n_out, int T, void* stream);
int pbsed_add_inplace(float* a, const float* b, size_t n, void* stream);
int pbsed_conv_bwd_data(const float* g, const float* wd_packed, const unsigned char* unpool_idx,
                        const int* seq_len, float* dz, const float* bx, const float* bmean,
                        const float* binvstd, const float* bscale, const float* bshift, int relu,
                        double* stats, int B, int Cin, int Cout, int F, int T, int KH, int KW, void* stream);
int pbsed_conv_bwd_weight(const float* x, const float* scale, const float* shift, int relu,
                          const int* seq_len, const float* g, const unsigned char* unpool_idx, float* dw,
                          float* db, int B, int Cin, int Cout, int F, int T, int KH, int KW, void* stream);
/* pbsed_conv_bwd_weight with the batch-norm backward of the NEXT layer's input norm (padertorch Normalization behind
 * pb_sed/experiments/weak_label_crnn/training.py:218-242) applied while dY is staged: dz = that norm's masked ReLU-backward
 * gradient (output of pbsed_conv_bwd_data), gx = this conv's raw output, coef [3][Cout * (per_cf ? Fo : 1)] from
 * pbsed_bn_bwd_coef, gseq = the norm's sequence lengths; dY = k1 dz + k2 gx + k3 is also written to gout for the data
 * gradient.  Replaces pbsed_bn_bwd + pbsed_conv_bwd_weight.  PBSED_E_UNSUPPORTED where no kernel has the loader
 * (pbsed_conv_bwd_weight_bng_supported returns 1 / 0). */
int pbsed_conv_bwd_weight_bng(const float* x, const float* scale, const float* shift, int relu, const int* seq_len,
                              const float* dz, const float* gx, const float* coef, int per_cf, const int* gseq, float* gout,
                              const unsigned char* unpool_idx, float* dw, float* db, int B, int Cin, int Cout, int F, int T,
                              int KH, int KW, void* stream);
int pbsed_conv_bwd_weight_bng_supported(int KH, int KW, int Cin, int Cout, int F, int T, int per_cf);
/* bf16-MFMA operands (rounded while staged), fp32 accumulation / gradients; < 32 channels: the fp32 kernels. */
int pbsed_conv_bwd_weight_bf16(const float* x, const float* scale, const float* shift, int relu,
                          const int* seq_len, const float* g, const unsigned char* unpool_idx, float* dw,
                          float* db, int B, int Cin, int Cout, int F, int T, int KH, int KW, void* stream);

/* bf16-MFMA variants (same contracts; activations stay fp32 in HBM, operands are converted while staging, fp32
 * accumulate).  nsplit = 1: plain bf16 compute (BASELINE.json config 3).  nsplit = 3: exact 3-way bf16 split of both
 * operands, 6 partial products -> fp32-class accuracy.  w_packed_bf16: uint16 [nsplit][KH*KW][out_padded][in_padded]. */
/* Refresh many packed copies in one launch (e.g. all layers of a model after an optimiser step).  descs: DEVICE array. */
typedef struct {
    const float* src;       /* [Cout, Cin, KH, KW] weights */
    float* dst;             /* packed copy (sizes from pbsed_conv_pack_dims / pbsed_conv_pack_dims_wino) */
    int Cout, Cin, KH, KW, InP, OutP;
    int mode;               /* 0 / 1: direct forward / data-gradient layout, 2 / 3: Winograd forward / data-gradient,
                             * 4 / 5: bf16 (nsplit 1) forward / data-gradient layout of pbsed_pack_conv_weights_bf16 (dst: uint16),
                             * 6 / 7: its three-part (nsplit 3) form, 8 / 9: pbsed_pack_conv_weights_winox3 forward / data gradient,
                             * 10 / 11: pbsed_pack_conv1d_weights_x3 forward / data gradient (KH = 1),
                             * 12: dst [Cin][Cout] = src^T of a [Cout][Cin] fp32 matrix (the W^T operands of pbsed_gru_stack_bwd*) */
    int pad_;
} pbsed_pack_desc;
int pbsed_pack_conv_weights_batched(const pbsed_pack_desc* descs /*device*/, int n, void* stream);
/* 3x3 convs with the time axis in the Winograd F(4,3) domain (csrc/conv_wino.hip): same tensors, fusions and results
 * (fp32 MFMA, fp32 accumulate; ~1e-6 relative transform rounding) as pbsed_conv_fwd / pbsed_conv_bwd_data with
 * KH = KW = 3, half the multiplications.  u_packed: [3][6][in_padded][out_padded] from pbsed_pack_conv_weights_wino. */
void pbsed_conv_pack_dims_wino(int Cin, int Cout, int dgrad, int* in_padded /*host*/, int* out_padded /*host*/);
int pbsed_pack_conv_weights_wino(const float* w, float* u_packed, int Cout, int Cin, int dgrad, void* stream);
int pbsed_conv_fwd_wino(const float* x, const float* u_packed, const float* bias, const float* scale,
                        const float* shift, int relu, const int* seq_len, float* y, unsigned char* pool_idx,
                        double* stats, int stats_per_cf, int B, int Cin, int Cout, int F, int T, int pool, void* stream);
int pbsed_conv_bwd_data_wino(const float* g, const float* ud_packed, const unsigned char* unpool_idx,
                             const int* seq_len, float* dz, const float* bx, const float* bmean, const float* binvstd,
                             const float* bscale, const float* bshift, int relu, double* stats, int B, int Cin,
                             int Cout, int F, int T, void* stream);
/* The same 3x3 Winograd F(4,3) convolutions on the bf16 MFMA with exact three-way bf16 splits of both transformed operands
 * (csrc/conv_winox3.hip; fp32-class results: six part products above 2^-24 accumulated in fp32).  Replaces the
 * MFMA-bound 3x3 Conv2d + Normalization + ReLU (+ pool) sites of pb_sed/models/weak_label/crnn.py:93 (config
 * pb_sed/experiments/weak_label_crnn/training.py:159-169,218-229).  u_packed_x3: uint16
 * [in_padded/32][6][3][out_padded/16][3 parts][64 lanes][8] from pbsed_pack_conv_weights_winox3 (pbsed_pack_desc modes 8 / 9). */
void pbsed_conv_pack_dims_winox3(int Cin, int Cout, int dgrad, int* in_padded /*host*/, int* out_padded /*host*/);
int pbsed_pack_conv_weights_winox3(const float* w, unsigned short* u_packed_x3, int Cout, int Cin, int dgrad, void* stream);
int pbsed_conv_fwd_winox3(const float* x, const unsigned short* u_packed_x3, const float* bias, const float* scale,
                          const float* shift, int relu, const int* seq_len, float* y, unsigned char* pool_idx,
                          double* stats, int stats_per_cf, int B, int Cin, int Cout, int F, int T, int pool, void* stream);
int pbsed_conv_bwd_data_winox3(const float* g, const unsigned short* ud_packed_x3, const unsigned char* unpool_idx,
                               const int* seq_len, float* dz, const float* bx, const float* bmean, const float* binvstd,
                               const float* bscale, const float* bshift, int relu, double* stats, int B, int Cin,
                               int Cout, int F, int T, void* stream);
/* The FEW-CHANNEL 3x3 layers of the fp32 path - contraction over <= 16 channels into <= 32: the 16->16 (+ pool) and 16->32
 * Conv2d + Normalization + ReLU sites of pb_sed/experiments/weak_label_crnn/training.py:161-168, the tag-conditioned 11->16 first
 * layer of pb_sed/models/strong_label/crnn.py:60-75 - on the bf16 MFMA with exact three-way bf16 operand splits
 * (csrc/conv_s16.hip; fp32-class results; K = 32 of an MFMA = two taps x 16 channels).  Same contracts as pbsed_conv_fwd /
 * pbsed_conv_bwd_data with KH = KW = 3.  wfrag: uint16 [5 tap pairs][3 parts][out_padded/16][64 lanes][8] from
 * pbsed_pack_conv_weights_s16 (pbsed_pack_desc modes 14 / 15); forward: Cin <= 16, Cout <= 32; data gradient: Cout <= 16 (the
 * contraction), Cin <= 32 (produced) - anything else returns PBSED_E_UNSUPPORTED. */
void pbsed_conv_pack_dims_s16(int Cin, int Cout, int dgrad, int* in_padded /*host*/, int* out_padded /*host*/);
int pbsed_pack_conv_weights_s16(const float* w, unsigned short* wfrag, int Cout, int Cin, int dgrad, void* stream);
int pbsed_conv_fwd_s16(const float* x, const unsigned short* wfrag, const float* bias, const float* scale,
                       const float* shift, int relu, const int* seq_len, float* y, unsigned char* pool_idx,
                       double* stats, int stats_per_cf, int B, int Cin, int Cout, int F, int T, int pool, void* stream);
int pbsed_conv_bwd_data_s16(const float* g, const unsigned short* wfrag_d, const unsigned char* unpool_idx,
                            const int* seq_len, float* dz, const float* bx, const float* bmean, const float* binvstd,
                            const float* bscale, const float* bshift, int relu, double* stats, int B, int Cin,
                            int Cout, int F, int T, void* stream);
/* Conv1d (kernel size 1 or 3, zero padding) forward / data gradient on [B, C, T] tensors with exact three-way bf16 operand
 * splits on the bf16 MFMA, producer / consumer form (csrc/conv1d_pc.hip; fp32-class results).  Replaces the CNN1d layers and
 * per-frame output nets of pb_sed/models/weak_label/crnn.py:93-101 and pb_sed/models/strong_label/crnn.py:88-104 in the
 * fp32 path: same prologue (BN-apply + ReLU + mask), bias, masked batch statistics [PBSED_STAT_SLOTS][Cout][2] and
 * BN-ReLU-backward epilogue as pbsed_conv_fwd / pbsed_conv_bwd_data with F = 1, KH = 1.  u_packed_x3: uint16
 * [in_padded/32][KW][out_padded/16][3 parts][64 lanes][8] from pbsed_pack_conv1d_weights_x3 (pbsed_pack_desc modes 10 / 11). */
void pbsed_conv1d_pack_dims_x3(int Cin, int Cout, int dgrad, int* in_padded /*host*/, int* out_padded /*host*/);
int pbsed_pack_conv1d_weights_x3(const float* w, unsigned short* u_packed_x3, int Cout, int Cin, int KW, int dgrad, void* stream);
int pbsed_conv1d_fwd_x3(const float* x, const unsigned short* u_packed_x3, const float* bias, const float* scale,
                        const float* shift, int relu, const int* seq_len, float* y, double* stats, int B, int Cin, int Cout,
                        int T, int KW, void* stream);
int pbsed_conv1d_bwd_data_x3(const float* g, const unsigned short* ud_packed_x3, const int* seq_len, float* dz, const float* bx,
                             const float* bmean, const float* binvstd, const float* bscale, const float* bshift, int relu,
                             double* stats, int B, int Cin, int Cout, int T, int KW, void* stream);
void pbsed_conv_pack_dims_bf16(int Cin, int Cout, int dgrad, int* in_padded /*host*/, int* out_padded /*host*/);
int pbsed_pack_conv_weights_bf16(const float* w, unsigned short* w_packed_bf16, int Cout, int Cin, int KH, int KW,
                                 int dgrad, int nsplit, void* stream);
int pbsed_conv_fwd_bf16(const float* x, const unsigned short* w_packed_bf16, const float* bias, const float* scale,
                        const float* shift, int relu, const int* seq_len, float* y, unsigned char* pool_idx,
                        double* stats, int stats_per_cf, int B, int Cin, int Cout, int F, int T, int KH, int KW, int pool,
                        int nsplit, void* stream);
/* pbsed_conv_fwd_bf16 whose (biased, pooled) output gets `residual` [B, Cout, Fo, T] added before the store and the
 * statistics (see pbsed_conv_fwd_res): the 1x1 conv2d layers of net_config 'deep'
 * (pb_sed/experiments/weak_label_crnn/training.py:170-183) with fp32-class bf16x3 operands (nsplit = 3). */
int pbsed_conv_fwd_bf16_res(const float* x, const unsigned short* w_packed_bf16, const float* bias, const float* scale,
                            const float* shift, int relu, const int* seq_len, float* y, unsigned char* pool_idx,
                            double* stats, int stats_per_cf, int B, int Cin, int Cout, int F, int T, int KH, int KW, int pool,
                            int nsplit, const float* residual, void* stream);
int pbsed_conv_bwd_data_bf16(const float* g, const unsigned short* wd_packed_bf16, const unsigned char* unpool_idx,
                             const int* seq_len, float* dz, const float* bx, const float* bmean, const float* binvstd,
                             const float* bscale, const float* bshift, int relu, double* stats, int B, int Cin, int Cout,
                             int F, int T, int KH, int KW, int nsplit, void* stream);

/* ---- Normalization ('batch', eps 1e-3; pb_sed/experiments/weak_label_crnn/training.py:223-225) */
int pbsed_bn_finalize(const double* sums, double count, const float* gamma, const float* beta, float eps,
                      float momentum, float* running_mean, float* running_power, float* mean, float* invstd,
                      float* scale, float* shift, int C, void* stream);
int pbsed_bn_eval_params(const float* gamma, const float* beta, float eps, const float* running_mean,
                         const float* running_power, float* mean, float* invstd, float* scale, float* shift,
                         int C, void* stream);
/* bn_bwd_finalize + bn_bwd_apply fused (one launch): dz [B,C,S,T] in place -> gradient wrt the BN input; sums =
 * [PBSED_STAT_SLOTS][C][2] partial (sum dz, sum dz*xhat) from pbsed_conv_bwd_data; dgamma / dbeta accumulate. */
int pbsed_bn_bwd(float* dz, const float* x, const double* sums, double count, const float* mean, const float* invstd,
                 const float* scale, float* dgamma, float* dbeta, const int* seq_len, int B, int C, int S, int T,
                 void* stream);
int pbsed_bn_bwd_finalize(const double* sums, double count, float* dgamma, float* dbeta, float* m1, float* m2,
                          int C, void* stream);
/* sums -> dgamma, dbeta (+=) and coef [3][C]: dx = coef[0][c] dz + coef[1][c] x + coef[2][c] (pbsed_conv_bwd_weight_bng). */
int pbsed_bn_bwd_coef(const double* sums, double count, const float* mean, const float* invstd, const float* scale,
                      float* dgamma, float* dbeta, float* coef, int C, void* stream);
int pbsed_bn_bwd_apply(float* dz, const float* x, const float* mean, const float* invstd, const float* scale,
                       const float* m1, const float* m2, const int* seq_len, int B, int C, int S, int T,
                       void* stream);

/* ---- GRU recurrence (torch.nn.GRU inside padertorch's GRU wrapper; pb_sed/models/weak_label/crnn.py:61-67,
 * pb_sed/models/strong_label/crnn.py:92).  Arrays of per-chain pointers are HOST arrays of device pointers. */
int pbsed_gru_scan_fwd(int nchains, const float* const* gi, const float* const* w_hh, const float* const* b_hh,
                       float* const* hs, float* const* save, const int* reverse /*host*/,
                       const int* seq_len, int B, int H, int T, void* stream);
int pbsed_gru_scan_bwd(int nchains, const float* const* w_hh_t, const float* const* hs,
                       const float* const* save, const float* const* dy, float* const* dgi, float* const* dgh,
                       float* const* dhz, const int* reverse /*host*/, const int* seq_len, int B, int H, int T,
                       void* stream);
/* Multi-layer UNIDIRECTIONAL stacks (FBCRNN: forward + time-reversed 2-layer GRUs) as a layer wavefront of
 * T + nlayers - 1 per-step launches (the fallback of the persistent *_granule scans below; `save` rows [4][H]).
 * Pointer tables are host arrays indexed [chain*nlayers + layer].  1 <= nchains <= 6 (the chains are independent: both
 * directions of up to three networks that share B, H, T and seq_len - ensemble inference runs its detectors' layers as one
 * launch), 1 <= nlayers <= 4; the same limits hold for the *_granule entry points. */
int pbsed_gru_stack_fwd(int nchains, int nlayers, const float* const* gi0, const float* const* w_ih,
                        const float* const* b_ih, const float* const* w_hh, const float* const* b_hh,
                        float* const* hs, float* const* save, const int* reverse /*host*/, const int* seq_len,
                        int B, int H, int T, void* stream);
/* Blocks of the persistent ("granule") scan kernel of this shape that can be co-resident on the CURRENT device: occupancy API
 * of that kernel instantiation x CUs, at most one block per CU.  A scan launches nchains * (2 nlayers - 1) * (H / 16) *
 * ceil(B / (16 * tiles_per_block)) blocks that poll each other's words; run the granule entry points only when that count is
 * within the capacity (the callers here keep it within 7/8 of it) and pbsed_gru_stack_fwd / _bwd (one launch per time step)
 * otherwise.  The granule entry points check the same bound and return PBSED_E_UNSUPPORTED rather than launch a scan that
 * could never finish. */
int pbsed_gru_granule_capacity(int H, int bwd, int bf16, int tiles_per_block);
/* First-poll delays of the persistent scans on the current device, in units of 64 clocks, per scan kind (0: two-layer stacks,
 * 1: one-layer scans, 2: forward with two batch tiles per block): {forward, forward gate waves, BPTT, BPTT gate waves}.  A wave
 * sleeps this long before its first poll of a time step (polling before the data can be there loads the fabric); the forward
 * optimum is sharp and depends on clocks and on what ran before the scan, so callers measure it in place (pb_sed_amd/ops.py
 * does at the first scan of every shape) instead of relying on the built-in defaults. */
int pbsed_gru_get_poll_delays(int kind, int* out4 /*host*/);
int pbsed_gru_set_poll_delays(int kind, int fwd, int fwd_gate, int bwd, int bwd_gate);
/* XCD-local exchange of the BPTT scans.  The last ring of a BPTT scan hands its states over through its XCD's L2 alone
 * (plain stores, a plain first look) - valid only when all workgroups of that ring run on ONE XCD.  The library verifies the
 * dispatcher's placement once per device with a probe launch (XCC_ID per workgroup; the exchange stays off on any device
 * where block ids equal mod 8 do not share an XCD, or fewer than 8 XCDs answer), and every workgroup of such a ring
 * re-checks its own XCC_ID at launch: a mismatch sets bit 1 (value 2) of the scan's err_flag (bit 0 = a hand-off timed
 * out).  on = 0 switches the exchange off for the process (callers do on seeing bit 1; PBSED_GRU_XCD_LOCAL=0 does the
 * same from the environment), on = 1 allows it again.  Returns the previous setting.  Replaces nothing in the reference
 * (torch.nn.GRU has no such knob: pb_sed/models/weak_label/crnn.py:61-67). */
int pbsed_gru_set_xcd_local(int on);
/* Diagnostics of the persistent scans: the scans launched after this call write shader-clock stamps of workgroup `block`
 * (1-D index of the launch), scan steps 200..231, to buf[32][16] (device, 4 KB) - slots 0..4: first contraction wave at the top
 * of the step / first poll issued / poll satisfied / partial sums written / past the barrier; 5: poll attempts that missed;
 * 8..12: first gate wave at the top / past the barrier / partial sums reduced / state published / outputs stored.
 * buf = NULL switches it off (the default).  tools/gru_scan_prof.py prints the timeline the rooflines of DESIGN.md quote. */
int pbsed_gru_set_prof(unsigned long long* buf, int block);
/* Persistent forward scan (one launch for the whole scan; needs nchains*(2*nlayers-1)*(H/16)*ceil(B/16) <= #CUs
 * co-resident workgroups) exchanging h_t (and the projected inputs of layers > 0) between workgroups as 4-byte words
 * = the fp32 value with its mantissa LSB replaced by the call's parity bit (the exchanged quantity is defined as the
 * truncated value).  granules: device uint32 workspace of nchains*T*Bp*H*(nlayers + 3*(nlayers-1)) words (Bp = B rounded up to 16:
 * the exchanged states are stored as [T][batch tile][H/16][16 rows][16 units] tiles), ZERO before
 * its first use; epoch: odd on the first use of a workspace, parity flipped on every further call with it.
 * err_flag: device uint32, non-zero afterwards if a hand-off timed out (re-zero the workspace then).
 * `save` rows are [5][H] (factors of dh_t, see gru_stack.hip).  When the one-tile-per-block grid exceeds 7/8 of the capacity
 * the forward scan gives every block two batch tiles, run half a step apart (one tile's states travel while the other is
 * computed). */
int pbsed_gru_stack_fwd_granule(int nchains, int nlayers, const float* const* gi0, const float* const* w_ih,
                                const float* const* b_ih, const float* const* w_hh, const float* const* b_hh,
                                float* const* hs, float* const* save, const int* reverse /*host*/, const int* seq_len,
                                int B, int H, int T, unsigned int* granules, unsigned int epoch,
                                unsigned int* err_flag, void* stream);
int pbsed_gru_stack_bwd(int nchains, int nlayers, const float* const* w_hh_t, const float* const* w_ih_up_t,
                        const float* const* hs, const float* const* save, const float* const* dy_top,
                        float* const* dgi, float* const* dgh, float* const* dhz, const int* reverse /*host*/,
                        const int* seq_len, int B, int H, int T, void* stream);
/* Persistent BPTT exchanging dh_t / dy_t the same way (gate gradients are rebuilt by the consumer from the factors
 * the granule forward scan saved).  granules: device uint32 workspace of nchains*T*Bp*H*(2*nlayers-1) words. */
int pbsed_gru_stack_bwd_granule(int nchains, int nlayers, const float* const* w_hh_t, const float* const* w_ih_up_t,
                                const float* const* hs, const float* const* save, const float* const* dy_top,
                                float* const* dgi, float* const* dgh, const int* reverse /*host*/, const int* seq_len,
                                int B, int H, int T, unsigned int* granules, unsigned int epoch,
                                unsigned int* err_flag, void* stream);
/* The same scans with plain bf16 operands of the recurrent / layer-boundary products (weights and h_{t-1} / gate gradients
 * rounded to bf16 for the MFMA; fp32 accumulation, state, gate maths and outputs): the bf16 training mode
 * (BASELINE.json configs[2]).  Same arguments, workspaces and save format as the fp32-class entry points. */
int pbsed_gru_stack_fwd_granule_bf16(int nchains, int nlayers, const float* const* gi0, const float* const* w_ih,
                                     const float* const* b_ih, const float* const* w_hh, const float* const* b_hh,
                                     float* const* hs, float* const* save, const int* reverse /*host*/, const int* seq_len,
                                     int B, int H, int T, unsigned int* granules, unsigned int epoch,
                                     unsigned int* err_flag, void* stream);
int pbsed_gru_stack_bwd_granule_bf16(int nchains, int nlayers, const float* const* w_hh_t, const float* const* w_ih_up_t,
                                     const float* const* hs, const float* const* save, const float* const* dy_top,
                                     float* const* dgi, float* const* dgh, const int* reverse /*host*/, const int* seq_len,
                                     int B, int H, int T, unsigned int* granules, unsigned int epoch,
                                     unsigned int* err_flag, void* stream);
int pbsed_bct_to_tbc(const float* src, float* dst, int B, int C, int T, void* stream);
int pbsed_tbc_to_bct(const float* src, float* dst, int B, int C, int T, int shift, void* stream);
int pbsed_transpose2d(const float* src, float* dst, int R, int C, void* stream);
/* Weight / bias gradients of n GRU matrices from the scans' time-major buffers (nn.GRU backward,
 * pb_sed/models/base.py:64-68): dw[i][g][k] += sum_{t,b} dg[i][t][b][g] * x[i][t+shift[i]][b][k] (rows outside
 * [0,T) are zero), db[i][g] += sum_{t,b} dg[i][t][b][g] (db or db[i] may be NULL).  dg: [T,B,G], x: [T,B,K];
 * pointer tables and shift are host arrays; G, K multiples of 4. */
int pbsed_gru_wgrad(int n, const float* const* dg, const float* const* x, const int* shift /*host*/,
                    float* const* dw, float* const* db, int T, int B, int G, int K, void* stream);
/* The fp32 entry point computes the products with exact three-way bf16 operand splits on the bf16 MFMA (fp32-class
 * results).  pbsed_gru_wgrad_multi: the same for GEMMs of different
 * width in ONE launch (K: host array of n input widths; dw[i] is [G, K[i]]) - all weight gradients of a GRU backward pass
 * together; bf16 != 0: plain bf16 operands with fp32 accumulation, the bf16 training mode (BASELINE.json configs[2]). */
int pbsed_gru_wgrad_multi(int n, const float* const* dg, const float* const* x, const int* shift /*host*/,
                          float* const* dw, float* const* db, int T, int B, int G, const int* K /*host*/, int bf16,
                          void* stream);

/* Time-major projections around the scans: y [R, N] = bias [N] (or 0 if NULL) + sum_i x[i] [R, k[i]] @ w[i] [N, k[i]]^T
 * with R = T*B rows in the scans' own [T][B][*] layout - the GRU input projection gi = W_ih x + b_ih of torch.nn.GRU
 * (pb_sed/models/weak_label/crnn.py:61-67) and its data gradient dx = sum over chains of dgi @ W_ih (pass W_ih^T as w),
 * without the [B,C,T] <-> [T,B,C] round trips of a convolution on the CNN layout.  x, w, k: host arrays of n_src <= 4
 * entries; N and every k[i] multiples of 4.  bf16 = 0: exact three-way bf16 operand splits on the bf16 MFMA (fp32-class
 * results); bf16 != 0: plain bf16 operands, fp32 accumulation. */
int pbsed_tm_gemm(int n_src, const float* const* x, const float* const* w, const int* k /*host*/, const float* bias,
                  float* y, int R, int N, int bf16, void* stream);

/* masked per-channel sums (sum x, sum x^2 over frames < seq_len) of a network input x [B, C, S, T] into stats
 * [PBSED_STAT_SLOTS][C][2] (zeroed doubles): batch statistics of a FIRST layer that has its own pre-activation norm
 * (padertorch CNN input_layer=False; every other norm gets its statistics from the producing convolution's epilogue). */
int pbsed_channel_stats(const float* x, const int* seq_len, double* stats, int B, int C, int S, int T, void* stream);
/* A norm + ReLU that CLOSES a stack (padertorch pre-activation CNN1d / CNN2d with a final norm + activation behind the last
 * conv; SURVEY.md A.4 variant (iii) of pb_sed/experiments/weak_label_crnn/training.py:218-242): y = mask * relu(x * scale[c] +
 * shift[c]) on [B, C, S, T]; backward writes dz = dy * mask * relu'(z) and the (sum dz, sum dz * xhat) partial sums
 * [PBSED_STAT_SLOTS][C][2] (zeroed doubles) that pbsed_bn_bwd then turns into the gradient wrt x, dgamma and dbeta. */
int pbsed_bn_relu_fwd(const float* x, const float* scale, const float* shift, const int* seq_len, float* y, int relu, int B,
                      int C, int S, int T, void* stream);
int pbsed_bn_relu_bwd(const float* dy, const float* x, const float* scale, const float* shift, const float* mean,
                      const float* invstd, const int* seq_len, float* dz, double* stats, int relu, int B, int C, int S, int T,
                      void* stream);

/* ---- heads' squash + losses (pb_sed/models/weak_label/crnn.py:58-59,107-206;
 * pb_sed/models/strong_label/crnn.py:93,106-112) */
int pbsed_squash_fwd(const float* x, float* y, size_t n, float eps, void* stream);
int pbsed_squash_bwd(const float* y, const float* dy, float* dx, size_t n, float eps, void* stream);
int pbsed_fbcrnn_loss(const float* logit_fwd, const float* logit_bwd, const float* weak_targets,
                      const float* boundary_targets, const float* class_weights, const int* seq_len,
                      float* y_fwd, float* y_bwd, float* dlogit_fwd, float* dlogit_bwd, float* loss, int B, int K,
                      int T, float minimum_score, float strong_weight, int slat, float label_smoothing,
                      int inputs_are_scores, float* summary /* optional [3*B*K + 1]: weak-label mask, masked weak targets,
                      clip-level scores y_fwd[L-1] (/2 + y_bwd[0]/2), boundary label rate (CRNN.review's host side) */,
                      void* stream);
int pbsed_bicrnn_loss(const float* logit, const float* strong_targets, const int* seq_len, float* y,
                      float* dlogit, float* loss, double* scratch, int B, int K, int T, int inputs_are_scores,
                      void* stream);

/* Validation summary of strong_label.CRNN.review (pb_sed/models/strong_label/crnn.py:114-136): y, strong_targets
 * [B,K,T] -> segment maxima over `segment_length` frames y_seg / t_seg [B, T/segment_length, K] (segments not fully
 * inside seq_len[b] hold 0), mask_mean [B,K] = share of labelled frames (target > .99 or < .01) inside the sequence,
 * mask_cnt [B,K] = labelled frames over all T. */
int pbsed_bicrnn_review_summary(const float* y, const float* strong_targets, const int* seq_len, float* y_seg, float* t_seg,
                                float* mask_mean, float* mask_cnt, int B, int K, int T, int segment_length, void* stream);

/* ---- ensemble post-processing (pb_sed/models/base/inference.py:142-184,225-289; pb_sed/filters.py:56-83,112-135;
 * event extraction = sed_scores_eval scores_to_event_list, call site experiments/strong_label_crnn/inference.py:147-150).
 * Rows = flattened leading dims of a [B,(n,)K,T] score tensor; n_row / thr / len are per-row device arrays. */
int pbsed_ensemble_mean_mask(const float* const* scores /*host array of device ptrs*/, int n_models, float* out,
                             const int* seq_len, int rows_per_clip, int R, int T, void* stream);
int pbsed_medfilt(const float* in, float* out, const int* n_row, int R, int T, void* stream);
int pbsed_boundariesfilt(const float* in, float* out, double* out64, const int* n_row, int R, int T, void* stream);
int pbsed_event_frames(const float* scores, const float* thr, const int* len, int* events, int* counts, int R, int T,
                       int max_events, void* stream);

/* ---- optimiser (padertorch Adam(lr, gradient_clipping); pb_sed/experiments/weak_label_crnn/training.py:264-269) */
int pbsed_grad_sumsq(const float* g, size_t n, double* out, void* stream);
/* skip_flags [n_flags] (or NULL): device error words of this step's persistent GRU scans (their `err` argument); if
 * any is non-zero the update is skipped on the device, so gradients of a timed-out scan are never applied. */
int pbsed_adam_step(float* p, const float* g, float* m, float* v, size_t n, float lr, float beta1, float beta2,
                    float eps, int step, float grad_scale, float max_norm, const double* sumsq, float* norm_out,
                    const int* skip_flags, int n_flags, void* stream);
int pbsed_memset_async(void* p, int value, size_t bytes, void* stream);

/* ---- data-parallel gradient exchange (new in the build: the reference is single-device,
 * pb_sed/experiments/weak_label_crnn/training.py:284).  RCCL over xGMI, one communicator per process /
 * GPU.  The communicator owns one extra HIP stream: pbsed_allreduce_begin orders an in-place fp32 sum-all-reduce of
 * buf[0..n) after everything already enqueued on `producer_stream` and runs it on the communicator's stream (it overlaps
 * what the producer stream does next); pbsed_allreduce_finish makes `consumer_stream` wait for every collective begun
 * since the last finish.  Nothing blocks the host.  The unique id (pbsed_comm_id_bytes() host bytes, made on rank 0) is
 * distributed by the caller. */
int pbsed_comm_id_bytes(void);
int pbsed_comm_unique_id(void* out /*host*/);
int pbsed_comm_create(const void* unique_id /*host*/, int rank, int world, void** comm /*host, out*/);
int pbsed_comm_destroy(void* comm);
int pbsed_allreduce_begin(void* comm, float* buf, size_t n, void* producer_stream);
int pbsed_allreduce_finish(void* comm, void* consumer_stream);

#ifdef __cplusplus
}
#endif
#endif /* PBSED_H */
