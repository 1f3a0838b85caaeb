/*
 * dd_hip.h -- C ABI of libdd_hip.so: hand-written CDNA4 (gfx950) HIP kernels for the conv hot
 * path of DeepBlender/DeepDenoiser.
 *
 * The reference has no FFI/plugin interface (it is pure Python on TensorFlow 1.x); its seams are
 * Python call signatures.  Each entry point below names the reference op-level seam it replaces
 * (file:line relative to /root/reference).  Conventions:
 *   - caller (PyTorch host) owns every buffer; the library never allocates or frees device memory
 *     and keeps no global state; all work is enqueued on the passed hipStream_t; no implicit sync;
 *   - activations are NHWC; a tensor is (pointer, ld) where ld = channel stride of one pixel in
 *     ELEMENTS (so concat buffers are written in place at a channel offset);
 *   - master weights / gradients / optimizer state are fp32 in TensorFlow variable layout:
 *     conv kernel HWIO [kh,kw,C_in,C_out]; transpose-conv kernel [kh,kw,C_out,C_in];
 *   - `dtype` selects the storage type of activations and packed weights: DD_F32 (parity path,
 *     exact-f32 MFMA), DD_BF16 (training throughput path, bf16 MFMA with fp32 accumulate) or
 *     DD_F16 (inference path of BASELINE cfg-5: fp16 MFMA, fp32 accumulate; range +-65504);
 *   - return 0 on success, negative dd_status otherwise; dd_last_error() gives the message of the
 *     calling thread's last failure.  No exceptions cross the ABI.
 */
#ifndef DD_HIP_H
#define DD_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* dd_stream; /* hipStream_t */

enum dd_dtype { DD_F32 = 0, DD_BF16 = 1, DD_F16 = 2 };

enum dd_status {
  DD_OK = 0,
  DD_ERR_INVALID = -1,  /* bad argument / unsupported shape */
  DD_ERR_LAUNCH = -2    /* HIP launch error */
};

/* flags of dd_conv_igemm / dd_conv_wgrad */
enum dd_conv_flags {
  DD_IN_RELU = 1,       /* operand is relu(x) (pre-activation convs: Tiramisu.py:33, MultiScalePrediction.py:87) */
  DD_OUT_RELU = 2,      /* y = relu(.) (tf.layers activation=tf.nn.relu) */
  DD_ACCUM = 4,         /* y += result (gradient accumulation for multi-consumer tensors) */
  DD_PIXSHUF = 8,       /* 2x2/s2 transpose-conv forward: n=(a,b,co) scattered to (2y+a,2x+b,co) */
  DD_GATHER2X2 = 16     /* 2x2/s2 gather: tap (a,b) reads input pixel (2y+a,2x+b) (transpose-conv dgrad/wgrad) */
};

const char* dd_version(void);
const char* dd_last_error(void);

/* ---- weight packing: fp32 master (TF layout) -> MFMA operand layout [taps][n_pad][k_pad] of `dtype`.
 * dst[t][n][k] = src[tsrc*s_tap + n*s_n + k*s_k], tsrc = tap_flip ? taps-1-t : t; zero padded. */
int dd_pack_weights(const float* src, void* dst, int dtype, int taps, int n, int k, int n_pad, int k_pad,
                    long s_tap, long s_n, long s_k, int tap_flip, dd_stream stream);

/* all layers in ONE launch: `table` is a device array of n records */
/* dst_ld / dst_tap_stride (elements; 0 = k_pad / n_pad * k_pad): the record fills the [n_pad][k_pad] corner of every tap of a WIDER image
 * ([taps][n_pad][dst_ld]) -- several records stack their reductions side by side in one operand (the gather-form data gradient of dd_conv3x3_ks) */
typedef struct { const float* src; void* dst; int taps, n, k, n_pad, k_pad, tap_flip; long s_tap, s_n, s_k; long dst_ld, dst_tap_stride; } dd_pack_desc;
int dd_pack_weights_batched(const dd_pack_desc* table, int n_layers, int dtype, dd_stream stream);

/* ---- implicit-GEMM convolution on MFMA (forward and data-gradient of every conv-like layer).
 * Replaces tf.layers.conv2d 3x3/1x1 SAME (UNet.py:29-31; Tiramisu.py:35-37,50-52,77-79;
 * Architecture.py:238-243; MultiScalePrediction.py:64-66,73-75,88-90), tf.layers.conv2d_transpose
 * 2x2/s2 (UNet.py:56-58) and, with flipped/transposed packed weights, their TF-autodiff input gradients.
 *   y[b,p,n] = epi( sum_{t,k} in(x)[b, map_t(p), k] * wp[t][n][k] + bias[n] + res[b,p,n] )
 * taps: 9 (3x3, pad 1), 1, or 4 with DD_GATHER2X2.  epi: optional relu, then *(mask>0), then +y if DD_ACCUM. */
typedef struct {
  const void* x; int ldx; int cin;     /* input [B,Hin,Win,*]; cin = valid channels (multiple of 16 bytes) */
  const void* wp; int k_pad; int n_pad;/* packed weights [taps][n_pad][k_pad] */
  const float* bias; int nbias;        /* fp32 bias and its length (nbias <= n), or NULL */
  const void* res; int ldres;          /* residual added before activation, or NULL */
  const void* mask; int ldmask;        /* y *= (mask > 0), or NULL (ReLU backward fused into the producer) */
  void* y; int ldy; int n;             /* output channels actually stored (n <= n_pad, n % 4 == 0) */
  int B, H, W;                         /* GEMM-row pixel grid: output grid, except DD_PIXSHUF where it is the input grid */
  int taps; int flags; int dtype;
} dd_conv_args;
int dd_conv_igemm(const dd_conv_args* a, dd_stream stream);
/* Number of dd_conv_igemm calls this process routed to the wide 1x1 GEMM kernel (csrc/dd_conv_pw.hip: taps == 1, 2-byte storage, more than
 * 64 output channels, no residual operand, at least DD_CONV_PW_MIN_PIXELS = 32 768 pixels; DD_CONV_PW=0 turns the route off).  Tests use it
 * to assert that the kernel they mean to check is the one that ran. */
long dd_conv_pw_count(void);
long dd_wgrad_pw_count(void);   /* ... and dd_conv_wgrad calls routed to its weight-gradient counterpart (taps == 1, m or n above 64 channels) */

/* ---- weight gradient on MFMA: out[t][m][n] += sum_{b,p} in(P)[b,map_t(p),m] * Q[b,p,n]   (fp32 atomics)
 * conv2d:            P = layer input x, Q = pre-activation output gradient -> out = dKernel HWIO
 * conv2d_transpose:  P = output gradient (fine grid, DD_GATHER2X2), Q = layer input -> out = dKernel [a,b,co,ci]
 * Replaces the TF-autodiff filter gradients behind AdamOptimizer.minimize (Training.py:701-702). */
typedef struct {
  const void* p; int ldp; int m;       /* P operand and its logical channel count (= out dim m) */
  const void* q; int ldq; int n;       /* Q operand and its logical channel count (= out dim n); channels up to the next
                                          multiple of 16 bytes must be readable and zero in both operands */
  float* out;                          /* [taps][m][n] fp32, accumulated atomically (zero it first) */
  float* bias_out; int bias_mode;      /* fused bias gradient (fp32 atomics): 0 none, 1 = column sums of Q -> bias_out[n] (conv2d),
                                          2 = column sums of P -> bias_out[m] (conv2d_transpose) */
  int B, H, W;                         /* grid of Q (the reduction pixels) */
  int taps; int flags; int dtype;
  int ksplit;                          /* number of reduction splits (0 = choose) */
  /* Stacked form (round 5; taps = 9, 2-byte storage): the weight gradients of ALL convs of a Tiramisu dense block (Tiramisu.py:26-41) in one
   * launch.  Conv j reads the prefix [0, stack_m0 + j * stack_width) of the block's buffer and appends stack_width channels, so the output
   * gradients of the stack_blocks convs are ONE contiguous channel range: Q = that range (n = stack_blocks * stack_width), P = the longest
   * prefix (m = stack_m0 + (stack_blocks - 1) * stack_width).  Column block j of the product, rows below its conv's own input width, is conv j's
   * gradient: stack_out[j] [taps][stack_m0 + j * stack_width][stack_width] and, with bias_mode = 1, stack_bias[j] [stack_width] (fp32 atomics);
   * out / bias_out are not used.  stack_blocks = 0: the plain form above. */
  int stack_blocks; int stack_width; int stack_m0;
  float* stack_out[8]; float* stack_bias[8];
} dd_wgrad_args;
int dd_conv_wgrad(const dd_wgrad_args* a, dd_stream stream);

/* ---- fused backward of a 3x3 SAME conv2d (stride 1): the data gradient (Conv2DBackpropInput + the ReluGrad of the layer's input) AND the
 * weight / bias gradients (Conv2DBackpropFilter, BiasAddGrad) that TensorFlow's autodiff emits for tf.layers.conv2d (Training.py:701-702 over
 * UNet.py:38-48), from ONE pass over dy and x (dd_conv_igemm + dd_conv_wgrad fetch each of them twice).  bf16 / f16 storage.  With a data
 * gradient: cout <= 64 in one launch (64 input channels per workgroup column), 65 <= cout <= 96 in one launch of the kernel that gives a
 * workgroup a 32-channel third of the input against all output channels (round 6: the U-Net's 64 x 64 level, UNet.py:25-36); wider layers run
 * as one launch per 64 output channels, the later ones accumulating into dx.
 *   dx[p][ci] = (use_mask ? x[p][ci] > 0 : 1) * sum_{t,co} wd[t][ci][co] * dy[p + off(t)][co]      (accumulate != 0: added to the existing dx)
 *   dw[t][ci][co] += sum_p x[p + off(t)][ci] * dy[p][co]     (TensorFlow kernel layout [3][3][cin][cout], fp32 atomics: zero it first)
 *   db[co] += sum_p dy[p][co]                                  (optional) */
typedef struct {
  const void* dy; int ld_dy; int cout; /* gradient of the layer's output [B,H,W,ld_dy]; channels up to the next multiple of 8 readable and zero */
  const void* x; int ld_x; int cin;    /* the layer's input [B,H,W,ld_x] (weight-gradient operand and ReLU mask of dx) */
  const void* wd; int n_pad; int k_pad;/* data-gradient weights as dd_pack_weights lays them out for dd_conv_igemm: [9][n_pad (ci)][k_pad (co)] */
  void* dx; int ld_dx;                 /* [B,H,W,ld_dx]; NULL (then wd may be NULL too, and cout may exceed 64): only dw / db are computed */
  float* dw; float* db;                /* db may be NULL */
  int B, H, W;
  int use_mask; int accumulate; int dtype;
} dd_conv_bwd_args;
int dd_conv3x3_bwd(const dd_conv_bwd_args* a, dd_stream stream);
/* the weight / bias gradients (dx = NULL) of n <= 4 layers on the SAME [B, H, W] grid as one launch (round 5: the 128-channel layers of the U-Net's
 * coarsest level -- each problem gets a share of the workgroups with more tiles apiece, i.e. fewer fp32 atomics per problem and n - 1 launch
 * boundaries fewer).  a: HOST array of n descriptors. */
int dd_conv3x3_bwd_multi(const dd_conv_bwd_args* a, int n, dd_stream stream);

/* ---- 2x2 / stride-2 transposed convolution (tf.layers.conv2d_transpose(filters, 2, strides=2), UNet.py:54-59) as streaming kernels: the
 * forward (+ bias, ReLU), and the data + weight + bias gradients of TensorFlow's autodiff (Training.py:701-702) in ONE launch.  bf16 / f16
 * storage, cin <= 128, cout a multiple of 16 (<= 96 forward, <= 128 backward: blocks of 64 run as consecutive launches).  x [B,H,W,ld_x], y / dy [B,2H,2W,ld_y];
 * kernel K[a][b][co][ci] (TensorFlow layout [2][2][cout][cin]):
 *   forward : y[2i+a][2j+b][co] = act(bias[co] + sum_ci x[i][j][ci] K[a][b][co][ci]);  w = dd_pack_weights layout [4*cout -> n_pad][k_pad (ci)]
 *   backward: dx[i][j][ci] = (use_mask ? x > 0 : 1) * sum_{a,b,co} dy[2i+a][2j+b][co] K[a][b][co][ci]   (accumulate: added to dx);
 *             dw[a][b][co][ci] += sum_{i,j} dy[2i+a][2j+b][co] x[i][j][ci];  db[co] += sum dy (db may be NULL); fp32 atomics: zero them first;
 *             w = the data-gradient pack [4][n_pad (ci)][k_pad (co)] */
typedef struct {
  const void* x; int ld_x; int cin;
  void* y; int ld_y; int cout;         /* forward: output; backward: dy (read only) */
  const void* w; int n_pad; int k_pad;
  const float* bias; int relu;         /* forward only */
  void* dx; int ld_dx; float* dw; float* db; int use_mask; int accumulate;      /* backward only */
  int B, H, W;                         /* INPUT grid */
  int dtype;
} dd_convt_args;
int dd_convt2x2_fwd(const dd_convt_args* a, dd_stream stream);
int dd_convt2x2_bwd(const dd_convt_args* a, dd_stream stream);

/* ---- K-streamed 3x3 convolution for deep reductions and few output channels: the dense blocks of the Tiramisu backbone
 * (tf.layers.conv2d over the growing concat, Tiramisu.py:26-41: K = 9 x up to 1 088 channels, 16 ... 128 new channels) and its 3x3 / stride-2
 * transposed convolution (tf.layers.conv2d_transpose(filters, 3, strides=2, padding='same'), Tiramisu.py:60-65).  bf16 / f16 storage.
 *   mode 0: y[p][n0 + c] = act(bias[n0 + c] + sum_{t,ci} w[t][n0 + c][ci] * in(x[p + off(t)][ci])),  c < n,  in = relu if DD_IN_RELU;
 *           w = the dd_pack_weights image [9][n_pad][k_pad] dd_conv_igemm takes, x [B,H,W,ldx], y [B,H,W,ldy] (a channel view of a concat buffer).
 *   mode 1..4: output parity (py, px) = ((mode-1)/2, (mode-1)%2) of the transposed conv (o = 2i + a, rows / columns 2H, 2W dropped):
 *           y[2i+py][2j+px][n0 + c] = act(bias + sum over the taps a = py, b = px (mod 2) of x[i - a/2][j - b/2][ci] K[a][b][n0 + c][ci]),
 *           y [B,2H,2W,ldy]; w = the image dd_pack_weights builds for the zero-stuffed form of the same layer ([9][n_pad][k_pad], taps flipped):
 *           the four parities together ARE the layer, with the 9 real taps instead of 36;
 *   mode 5: all four parities in one launch.
 * One call covers output channels [n0, n0 + n) (mode 5: n <= 64 + 63): whole blocks of 64 channels and the remainder (32 / 16-channel tiles: a
 * 96-channel layer = 64 + 32), and the parities, run as sub-problems of ONE grid (blockIdx.y); the input is re-read per block, from L2.
 * DD_ACCUM (mode 0) is the data gradient of a Tiramisu dense block in GATHER form (TF autodiff of Tiramisu.py:26-41 behind Training.py:701-702):
 * the gradient of a channel range of the block's buffer receives the contributions of ALL later convs of the block in one launch -- their output
 * gradients are one contiguous channel range of the gradient buffer (= x, the reduction), wp the stacked transposed / flipped kernels
 * (dd_pack_weights_batched with dst_ld) -- masked by the ReLU the consumers apply on read, added to what is already stored, rounded once. */
typedef struct {
  const void* x; int ldx; int cin;
  const void* wp; int n_pad; int k_pad;
  const float* bias; int nbias;        /* bias[c] for c < nbias; NULL: none */
  void* y; int ldy;
  int n0; int n;
  int B, H, W;                         /* INPUT grid */
  int mode; int flags; int dtype;      /* flags: DD_IN_RELU (mode 0 only) | DD_OUT_RELU, or DD_ACCUM alone (gather-form data gradient, below) */
  const void* mask; int ldmask;        /* DD_ACCUM: y[p][n0+c] += (mask[p][n0+c] > 0) * sum  -- no bias, no activation, rounded once */
} dd_conv_ks_args;
int dd_conv3x3_ks(const dd_conv_ks_args* a, dd_stream stream);

/* column sums: out[c] += sum_rows x[row*ld + c]  (bias gradients; embedding-row gradients) */
int dd_colsum(const void* x, int ld, int c, long rows, float* out, int dtype, dd_stream stream);
/* the same over n_segments (<= 64) consecutive segments of rows_per_segment rows in one launch: out[out_row[s]*out_ld + ch] += the sum over
 * segment s (the embedding-row gradients of FeatureFlags.feature_flags, FeatureFlags.py:57-67: one row per tuple, tiled over the tuple's images).
 * out_row is a HOST table; c <= 8 16-byte vectors of the storage type (64 bf16 / fp16, 32 f32 channels), x 16-byte aligned, ld a multiple of a vector. */
int dd_colsum_segments(const void* x, int ld, int c, long rows_per_segment, int n_segments, float* out, int out_ld, const int* out_row,
                       int dtype, dd_stream stream);

/* ---- max pooling, TF SAME (UNet.py:42-44 3x3/s2; Tiramisu.py:55-57 2x2/s2); idx = window argmax (uint8).
 * relu_mask != 0: x is a ReLU output whose backward mask (x > 0) is folded into idx (255 = the window maximum is not positive, no
 * gradient), so dd_maxpool_bwd can be called with mask == NULL.  idx == NULL: forward only (inference), no argmax plane is stored. */
int dd_maxpool_fwd(const void* x, int ldx, void* y, int ldy, uint8_t* idx, int C, int B, int H, int W,
                   int pool, int stride, int relu_mask, int dtype, dd_stream stream);
/* dx (+)= scatter(dy) masked by (mask>0) if mask != NULL */
int dd_maxpool_bwd(const void* dy, int lddy, const uint8_t* idx, void* dx, int lddx, const void* mask, int ldmask,
                   int C, int B, int H, int W, int pool, int stride, int accumulate, int dtype, dd_stream stream);

/* ---- f x f average pooling, stride f, fp32 (MultiScalePrediction.scale_down, MultiScalePrediction.py:11-13) */
int dd_avgpool(const float* x, int ldx, float* y, int ldy, int C, int B, int H, int W, int f, dd_stream stream);

/* ---- input pipeline (FeatureStandardization.standardize Architecture.py:39-46; FeatureEngineering.variance
 * FeatureEngineering.py:57-70): src [B,H,W,cs] fp32 (cs in {1,3}) -> dst [B,H,W,ldd] fp32 =
 * {3 standardized channels (1-ch sources replicated, SourceEncoder.py:49-51), local variance (1 channel, or cs if not compressed)}. */
typedef struct {
  int use_log1p; float mean; float inv_std;      /* standardize */
  int use_variance; int variance_before; int mode_neighbor; int relative; int compress; float epsilon;
} dd_feature_params;
int dd_prepare_feature(const float* src, int cs, float* dst, int ldd, const dd_feature_params* fp, int B, int H, int W, dd_stream stream);

/* channel gather/concat into the network input (SourceEncoder.prepare_neural_network_input, SourceEncoder.py:29-79,
 * incl. the embedding broadcast of FeatureFlags.feature_flags, FeatureFlags.py:50-69).
 * table: n_tuples x n_entries device records; dst [n_tuples*B,H,W,ld] of dtype, channels >= used ones zeroed up to c_pad. */
typedef struct { const float* src; int pixel_stride; int batch_stride_pixels; int nch; int dst_ch; } dd_gather_entry;
int dd_gather_input(const dd_gather_entry* table, int n_tuples, int n_entries, void* dst, int ld, int c_pad,
                    int B, int H, int W, int dtype, dd_stream stream);

/* prepare + gather in ONE launch (bf16 / fp16 / f32): every entry of a tuple is computed from the raw pass and written straight into the
 * network input; no fp32 staging plane.  kind 0: a render pass (src [B,H,W,cs] fp32, standardised per `fp`, local variance appended: nch = 3 +
 * variance channels); kind 1: a vector of nch floats broadcast over the image (the embedding row of FeatureFlags.feature_flags,
 * FeatureFlags.py:50-69); kind 2: a plane [B,H,W,nch] copied as it is (one-hot flags).  std_out (kind 0, optional): the entry's standardised
 * channels in fp32 [B,H,W,ld_std] -- the source KernelPredictor filters (Architecture.py:262-265). */
typedef struct { const float* src; int cs; int kind; dd_feature_params fp; int nch; int dst_ch; float* std_out; int ld_std; } dd_assemble_entry;
int dd_assemble_input(const dd_assemble_entry* table, int n_tuples, int n_entries, void* dst, int ld, int c_pad,
                      int B, int H, int W, int dtype, dd_stream stream);
/* The same launch for full-frame inference (Prediction.py:380-427: every tile is an H x W window of the frame): the kind-0 entries' src are
 * whole frames [frame_h, frame_w, cs] fp32 and image b of the batch is the window at (origins_yx[2b], origins_yx[2b+1]) (device table,
 * [B][2] int32, the tile plan's origins) -- read in place, mirrored at the WINDOW's border exactly as a copied tile would be.  Replaces one
 * dd_extract_tiles per render pass and the tile copies it wrote.  kind-1 / kind-2 entries as in dd_assemble_input. */
int dd_assemble_input_frames(const dd_assemble_entry* table, int n_tuples, int n_entries, void* dst, int ld, int c_pad,
                             int B, int H, int W, int dtype, const int* origins_yx, int frame_h, int frame_w, dd_stream stream);

/* ---- kernel prediction (KernelPrediction.kernel_prediction, KernelPrediction.py:11-63): softmax over k*k logits,
 * symmetric pad, per-pixel k x k filter of the 3-channel source. */
int dd_kpcn_fwd(const float* src, int ldsrc, const void* logits, int ldl, float* out, int ldo,
                int B, int H, int W, int ksize, int dtype, dd_stream stream);
int dd_kpcn_bwd(const float* src, int ldsrc, const void* logits, int ldl, const float* dout, int lddo,
                void* dlogits, int lddl, int dl_pad, int B, int H, int W, int ksize, int dtype, dd_stream stream);
/* The same with the logits computed IN the kernel, in fp32, from the input of AdjustNumberOfChannels' second 1x1 layer (Architecture.py:237-244):
 * logits[t] = bb[t] + sum_k hid[k] * wb[k * ldw + t], hid [B,H,W,ldh] of dtype (kh channels, post-ReLU), wb / bb the fp32 master variables
 * (TensorFlow layout [kh][ldw]; a COMBINED tuple's member j passes wb + j*k*k, bb + j*k*k).  What the layer-wise head of the half-precision
 * programs uses: a logit of magnitude 50 rounded to bf16 moves its softmax weight by up to 13 %.  The backward writes d logits (dtype) for the
 * weight / data gradient launches of that layer, exactly as dd_kpcn_bwd does. */
int dd_kpcn_hidden_fwd(const float* src, int ldsrc, const void* hid, int ldh, int kh, const float* wb, int ldw, const float* bb,
                       float* out, int ldo, int B, int H, int W, int ksize, int dtype, dd_stream stream);
int dd_kpcn_hidden_bwd(const float* src, int ldsrc, const void* hid, int ldh, int kh, const float* wb, int ldw, const float* bb,
                       const float* dout, int lddo, void* dlogits, int lddl, int dl_pad, int B, int H, int W, int ksize, int dtype, dd_stream stream);

/* ---- the whole kernel-prediction head of one scale as ONE launch each way: AdjustNumberOfChannels.predict (Architecture.py:237-244:
 * 1x1 conv C -> K + ReLU, 1x1 conv K -> K) + KernelPredictor.predict / KernelPrediction.kernel_prediction (Architecture.py:260-289,
 * KernelPrediction.py:11-63) with K = ksize * ksize (one tuple member), and their TF-autodiff gradients.  bf16 / fp16 storage only.
 * Weights are the fp32 master variables in TensorFlow layout (wa [C][K], wb [K][K]).  The backward recomputes hid / logits from x, so
 * the forward stores nothing; dx (storage type) is masked by x > 0 (x is a ReLU output) and optionally accumulated into; the weight /
 * bias gradients are ADDED to their fp32 arena slots. */
typedef struct {
  const void* x; int ldx; int C;           /* backbone output of this scale [N,H,W,ldx], C valid channels (C % 8 == 0, C <= 128) */
  const float* src; int ldsrc;             /* 3-channel source at this scale [N,H,W,ldsrc] fp32 */
  const float* wa; const float* ba; const float* wb; const float* bb;
  float* out; int ld_out;                  /* forward: prediction [N,H,W,ld_out] fp32 (3 channels) */
  const float* dout; int ld_dout;          /* backward: dL/d(out) fp32 */
  void* dx; int ld_dx; int accumulate_dx;  /* backward: dL/dx (storage type) */
  float* dwa; float* dba; float* dwb; float* dbb;
  int N, H, W, ksize, dtype;
} dd_head_args;
int dd_kpcn_head_fwd(const dd_head_args* a, dd_stream stream);
int dd_kpcn_head_bwd(const dd_head_args* a, dd_stream stream);
/* the backward of n <= 3 scales' heads (HOST array; one storage type and kernel size) as ONE launch: the scales are independent and the coarse
 * ones are mostly fixed cost (weight staging, the flush); side by side they share the device in proportion to their pixel counts (round 5) */
int dd_kpcn_head_bwd_multi(const dd_head_args* a, int n, dd_stream stream);

/* ---- multiscale compose (MultiScalePrediction.compose_scales, MultiScalePrediction.py:36-54) */
/* net input = concat(nearest_x2(small), fine), zero padded to c_pad channels */
int dd_compose_pack(const float* small, int lds, const float* fine, int ldf, void* dst, int ld, int c_pad,
                    int B, int H, int W, int dtype, dd_stream stream);
/* out = fine - w*up(avg2(fine)) + w*up(small), w = sigmoid(wl)  (wl = relu'd 1-channel net output) */
int dd_compose_blend_fwd(const float* small, int lds, const float* fine, int ldf, const void* wl, int ldw,
                         float* out, int ldo, int B, int H, int W, int dtype, dd_stream stream);
/* backward of the blend: dfine (overwritten), dsmall (overwritten or accumulated) and dwl (gradient at the pre-relu net output) */
int dd_compose_blend_bwd(const float* dout, int lddo, const float* small, int lds, const float* fine, int ldf,
                         const void* wl, int ldw, float* dsmall, int ldds, int accumulate_small, float* dfine, int lddf,
                         void* dwl, int lddw, int dw_pad, int B, int H, int W, int dtype, dd_stream stream);
/* backward of compose_pack: dsmall += sum_2x2 dnet[0:3]; dfine += dnet[3:6] */
int dd_compose_unpack_bwd(const void* dnet, int ld, float* dsmall, int ldds, float* dfine, int lddf,
                          int B, int H, int W, int dtype, dd_stream stream);

/* ---- the whole compose net of one scale transition as ONE launch (MultiScalePrediction.compose_scales, MultiScalePrediction.py:36-93):
 * out = fine - w * up(avg_pool2(fine)) + w * up(small),  w = sigmoid(relu(1x1(a3))),  a3 = two residual blocks over relu(1x1([up(small) | fine])).
 * Weights are the fp32 master variables of scope 'reused_compose_scales' in TensorFlow layout (conv2d: [6][24]; conv2d_1..4: HWIO
 * [3][3][24][24]; conv2d_5: [24][1]).  bf16 / fp16 storage only (f32 runs layer by layer through dd_conv_igemm).
 * Training: the activations the layer-wise backward reads are stored when their pointers are non-NULL -- netin (the packed 6-channel
 * input, ld >= 8), act[0..4] = a1, relu(r1), a2, relu(r3), a3 (24 channels, ld >= 24, ld % 8 == 0), wl = relu(1x1(a3)) (1 channel). */
typedef struct {
  const float* small; int ld_small;        /* [N, H/2, W/2, ld] fp32 */
  const float* fine; int ld_fine;          /* [N, H, W, ld] fp32 */
  float* out; int ld_out;                  /* [N, H, W, ld] fp32 */
  const float* w_in; const float* b_in;
  const float* w_res[4]; const float* b_res[4];
  const float* w_out; const float* b_out;
  void* save_netin; int ld_netin;
  void* save_act[5]; int ld_act[5];
  void* save_wl; int ld_wl;
  int N, H, W, dtype;
} dd_compose_args;
int dd_compose_net_fwd(const dd_compose_args* a, dd_stream stream);
/* Geometry dd_compose_net_fwd's row-streaming kernel (csrc/dd_compose_stream.hip) uses for an [N, H, W] launch on `cus` compute units:
 * out8 = {frame width, rows per step, 32-pixel tasks per row, column strips, strip output width, band height, bands per image,
 * virtual rows per band}.  Host-only (no device work); for tests and tools. */
int dd_compose_stream_plan(int N, int H, int W, int cus, int* out8);

/* Backward of dd_compose_net_fwd in ONE launch (TF autodiff of the same lines behind Training.py:701-702): d_fine is written, d_small is
 * written or accumulated into, every weight / bias gradient is ADDED to its fp32 arena slot (TensorFlow variable layout) with one
 * atomic per element per workgroup.  act[0..4], wl: the tensors the forward stored.  dout: dL/d(out), fp32. */
typedef struct {
  const float* small; int ld_small;
  const float* fine; int ld_fine;
  const float* dout; int ld_dout;
  const void* act[5]; int ld_act[5];
  const void* wl; int ld_wl;
  const float* w_in; const float* w_res[4]; const float* w_out;
  float* d_small; int ld_dsmall; int accumulate_small;
  float* d_fine; int ld_dfine;
  float* dw_in; float* db_in; float* dw_res[4]; float* db_res[4]; float* dw_out; float* db_out;
  int N, H, W, dtype;
  void* scratch; long scratch_bytes;       /* optional: >= dd_compose_bwd_scratch_bytes(N, H, W) bytes, 16-byte aligned (see below) */
} dd_compose_bwd_args;
int dd_compose_net_bwd(const dd_compose_bwd_args* a, dd_stream stream);
/* With `scratch` the backward runs as two row-streaming launches (csrc/dd_compose_stream_bwd.hip): the data-gradient chain, which parks the
 * four 24-channel gradient tensors and dz6 in scratch, then the weight gradients, which read every stored activation and parked gradient once.
 * Needs act[] rows of exactly 24 channels (ld_act == 24).  Without scratch (NULL): the 16x16-tile kernel of csrc/dd_compose.hip. */
long dd_compose_bwd_scratch_bytes(int N, int H, int W);

/* ---- inverse standardization (Architecture.py:48-55, Utilities.py:6-7), in place capable */
int dd_invert_std_fwd(const float* x, float* y, long n, int use_log1p, float mean, float std, dd_stream stream);
/* dx = dy * d(invert)/dx evaluated at the standardized value x */
int dd_invert_std_bwd(const float* x, const float* dy, float* dx, long n, int use_log1p, float mean, float std, dd_stream stream);

/* ---- loss head (LossDifference.difference LossDifference.py:15-35; BaseFeatureTraining.mean / variation_mean / loss
 * Training.py:126-129,141-176,210-243,304-348; Combined*FeatureTraining.initialize Training.py:420-437,475-495): per-pixel evaluation of
 * every feature loss, combined-feature loss (color*(direct+indirect)) and combined-image loss at one scale, fused with its own backward.
 * Each of the three levels has a MEAN term (difference of the values) and a VARIATION term (difference of the horizontal and vertical
 * finite differences x[.,j+1]-x[.,j], x[i+1,.]-x[i,.], averaged over all B*(H*(W-1)+(H-1)*W) pairs). */
#define DD_MAX_FEATURES 32
#define DD_MAX_COMBINED 8
typedef struct {
  int n_features;
  const float* pred[DD_MAX_FEATURES];    /* [B,H,W,pred_ld] internal prediction (1-channel passes use channel 0) */
  const float* target[DD_MAX_FEATURES];  /* [B,H,W,target_ld], first nch channels used */
  float* dpred[DD_MAX_FEATURES];         /* [B,H,W,3] overwritten (all 3 channels) */
  int target_ld[DD_MAX_FEATURES];        /* pixel stride of target */
  int pred_ld[DD_MAX_FEATURES];          /* pixel stride of pred (3, or 4 when a source is echoed) */
  int nch[DD_MAX_FEATURES];
  float weight[DD_MAX_FEATURES];         /* mean-term weight incl. scale factor; the 1/(B*H*W) mean is applied inside */
  float var_weight[DD_MAX_FEATURES];     /* variation-term weight incl. scale factor */
  float masked_weight[DD_MAX_FEATURES];  /* masked-mean weight incl. scale factor (BaseFeatureTraining.masked_mean, Training.py:131-137) */
  int mask_feature[DD_MAX_FEATURES];     /* feature whose TARGET defines the mask (Conv2dUtilities.non_zero_mask of the corresponding colour pass) */
  int n_combined;
  int comb[DD_MAX_COMBINED][3];          /* feature indices of color, direct, indirect */
  float comb_weight[DD_MAX_COMBINED];
  float comb_var_weight[DD_MAX_COMBINED];
  float comb_masked_weight[DD_MAX_COMBINED];
  int comb_mask_feature[DD_MAX_COMBINED];
  int n_image_combined; int image_combined[DD_MAX_COMBINED];   /* indices into comb[] */
  int n_image_features; int image_features[DD_MAX_FEATURES];   /* indices into features */
  float image_weight;
  float image_var_weight;
  int kind;                              /* 1 DIFFERENCE, 2 ABSOLUTE, 3 SMOOTH_ABSOLUTE, 4 SQUARED, 5 SMAPE */
  float epsilon;
  const float* mask_sums;                /* device, [DD_MAX_FEATURES + DD_MAX_COMBINED]: sum of each source's mask over the batch (dd_loss_mask_sums);
                                            may be NULL when no masked weight is set */
  /* Optional fusion of FeaturePrediction.prediction_invert_standardization (Architecture.py:134-138, :47-55; Utilities.py:3-7) into the loss
   * launch, per feature; for every descriptor WITHOUT variation terms (features-only ones take the flat-stream kernel, descriptors with
   * combined / image / masked terms the per-pixel kernel; with a variation term dd_loss_head rejects pred_std: run dd_invert_std_fwd / _bwd
   * around it).  With pred_std[f] != NULL the launch reads the STANDARDIZED prediction x from
   * pred_std[f], stores p = sign(z) expm1|z| (inv_log1p) or z, z = x inv_std + inv_mean, to pred_inv[f] (what dd_invert_std_fwd stores),
   * and writes dL/dx (what dd_invert_std_bwd stores) to dpred[f] instead of dL/dp; pred[f] is not read. */
  const float* pred_std[DD_MAX_FEATURES];
  float* pred_inv[DD_MAX_FEATURES];
  int inv_log1p[DD_MAX_FEATURES];
  float inv_mean[DD_MAX_FEATURES];
  float inv_std[DD_MAX_FEATURES];
} dd_loss_desc;
/* mask_sums[src] = sum over [B,H,W] of sign(sum_c |target_c|) of the source's mask feature, for every source with a masked weight
 * (features first, then combined at DD_MAX_FEATURES + k).  The masked mean divides by this BATCH-global count (Training.py:131-137). */
int dd_loss_mask_sums(const dd_loss_desc* desc, int B, int H, int W, float* mask_sums, dd_stream stream);
/* loss_out[0] += total weighted loss of this scale; dpred written. desc is a HOST struct (copied by value). */
int dd_loss_head(const dd_loss_desc* desc, int B, int H, int W, float* loss_out, float grad_scale, dd_stream stream);

/* ---- Adam, TensorFlow formulation (tf.train.AdamOptimizer, Training.py:701-702), flat arenas */
int dd_adam_step(float* params, const float* grads, float* m, float* v, long n, float lr_t, float beta1, float beta2,
                 float eps, float grad_scale, dd_stream stream);

/* ---- inference stitch (Prediction.py:384-441): copy crop windows of row-major tiles into the frame */
typedef struct { int tile; int crop_y0, crop_y1, crop_x0, crop_x1; int dst_img, dst_y, dst_x; } dd_stitch_entry;
/* frames: [n_img, frame_h, frame_w, ldf]; entry copies tiles[tile][crop] to frames[dst_img] at (dst_y, dst_x) */
int dd_stitch(const float* tiles, int tile_size, int ldt, float* frames, int frame_h, int frame_w, int ldf, int C,
              const dd_stitch_entry* table, int n_entries, dd_stream stream);

/* ---- inference tile extraction (Prediction.py:283-310: tiled_feature = feature[lower_h:upper_h, lower_w:upper_w] for every window
 * of the plan): tiles[i] = frame[origin_y[i] : +tile_size, origin_x[i] : +tile_size, 0:C].  frame [frame_h, frame_w, ldf],
 * tiles [n_tiles, tile_size, tile_size, ldt] fp32; origins: n_tiles (y, x) int pairs on the device. */
int dd_extract_tiles(const float* frame, int frame_h, int frame_w, int ldf, int C, float* tiles, int tile_size, int ldt,
                     const int* origins_yx, int n_tiles, dd_stream stream);

/* ---- recombination (Prediction.py:443-481): image = sum_k color_k*(direct_k+indirect_k) + sum_j single_j ; also writes each
 * combined_k if combined[k] != NULL.  All tensors [npix, 3] fp32 contiguous. */
typedef struct {
  int n_triples; const float* color[4]; const float* direct[4]; const float* indirect[4]; float* combined[4];
  int n_singles; const float* single[8];
  float* image;
} dd_recombine_desc;
int dd_recombine(const dd_recombine_desc* desc, long npix, dd_stream stream);

/* ---- small helpers of the graph executor */
/* dst (+)= src * (mask > 0)   (identity/residual gradient paths into ReLU outputs); mask may be NULL */
int dd_masked_add(void* dst, int lddst, const void* src, int ldsrc, const void* mask, int ldmask, int C, long npix,
                  int accumulate, int dtype, dd_stream stream);
/* dst[pix][0:nch] = convert(src[pix][0:nch]); dst[pix][nch:dst_pad] = 0   (dtype codes per tensor; fp32 <-> graph dtype) */
int dd_convert_channels(const void* src, int src_dtype, int ldsrc, void* dst, int dst_dtype, int lddst, int nch, int dst_pad,
                        long npix, dd_stream stream);
/* ---- data augmentation of one render pass of a batch of tiles (DataAugmentation.py:10-200 as applied by Training.py:794-821,
 * FeatureTrainingAugmentation Training.py:551-604): dst[b] = rotate_normal(permute_rgb(rotate_90(flip_left_right(src[b])))) with
 * per-tile draws.  src/dst: [B,H,W,C] float32, C = 1 or 3.  draws: B device records.  kind: DD_AUG_PLAIN (geometry only), DD_AUG_RGB
 * (RenderPasses.is_rgb_color_render_pass: the channel permutation applies), DD_AUG_NORMAL (world-space normal: the 3x3 rotation applies;
 * v_out = v_in * M, row vector times matrix), DD_AUG_SCREEN_NORMAL (flip negates x; rot90 k maps (x,y) -> (-y,x), (-x,-y), (y,-x)).
 * use_* : the DataAugmentationUsage switches.  Rotation by an odd k needs H == W (tiles are square). */
typedef struct { int flip; int rotate; int permute; float normal_rotation[9]; } dd_augment_draw;
enum { DD_AUG_PLAIN = 0, DD_AUG_RGB = 1, DD_AUG_NORMAL = 2, DD_AUG_SCREEN_NORMAL = 3 };
int dd_augment(const float* src, float* dst, int C, int B, int H, int W, const dd_augment_draw* draws, int kind,
               int use_flip, int use_rotate, int use_permute, int use_normal_rotation, dd_stream stream);

/* 3x3/s2 transpose conv (Tiramisu.py:62-64) = 3x3 SAME conv of the zero-stuffed input with the flipped kernel:
 * y[B,2H,2W,C] = 0 except y[2i+1,2j+1] = x[i,j]; and its adjoint dx[i,j] (+)= dy[2i+1,2j+1] * (mask>0). */
int dd_zero_stuff(const void* x, int ldx, void* y, int ldy, int C, int B, int H, int W, int dtype, dd_stream stream);
int dd_zero_unstuff(const void* dy, int lddy, void* dx, int lddx, const void* mask, int ldmask, int C, int B, int H, int W,
                    int accumulate, int dtype, dd_stream stream);

/* ---- backward of the 3x3/s2 transposed conv WITHOUT the zero-stuffed form (bf16 / f16 storage).  TensorFlow differentiates
 * tf.layers.conv2d_transpose (Tiramisu.py:60-65) into a stride-2 conv of the output gradient (the data gradient) and a filter gradient over the
 * output grid (Training.py:701-702).  With dy rearranged by output parity,
 *   s[b][i][j][(py*2 + px)*cp + co] = dy[b][2i + py][2j + px][co]        (dd_space_to_depth2; cp = cout rounded up to 16, the padding zero),
 * both become dense work on the INPUT grid:
 *   dx[i][j][ci]     = sum over the 9 taps (a, b) of  s[i + (a == 2)][j + (b == 2)][plane(a, b)*cp + co] * K[a][b][co][ci],  plane = (a & 1)*2 + (b & 1)
 *                      = dd_conv3x3_ks mode 6 over s (a 2 x 2-tap conv with offsets 0 / +1: 16 cout MACs per input pixel and ci against the 36 of
 *                      the 3x3 conv over the zero-stuffed 2H x 2W image, and no stuffed tensors);
 *   dK[a][b][co][ci] += sum_p s[p + off(a, b)][plane(a, b)*cp + co] * x[p][ci]      (dd_convt3_wgrad: exactly the 9 cout cin products per pixel).
 * TensorFlow kernel layout [3][3][cout][cin], fp32 atomics: zero dK first. */
int dd_space_to_depth2(const void* y, int ldy, void* s, int lds, int C, int cp, int B, int H, int W, int dtype, dd_stream stream);
typedef struct {
  const void* s; int lds; int cout; int cp;   /* space-to-depth output gradient [B,H,W,lds], lds >= 4 cp, cp % 16 == 0 */
  const void* x; int ldx; int cin;            /* the layer's input [B,H,W,ldx]; channels up to the next multiple of 8 readable and zero */
  float* dk;                                  /* [3][3][cout][cin] fp32 */
  int B, H, W;                                /* input grid */
  int dtype;
} dd_convt3_wgrad_args;
int dd_convt3_wgrad(const dd_convt3_wgrad_args* a, dd_stream stream);

/* ---- host-side helper (no device work): CRC-32C (Castagnoli) of `n` bytes, continuing from `crc` (0 to start).  The checksum of
 * the reference's on-disk formats: TFRecord framing (TFRecordsCreator.py:233-252) and TensorFlow checkpoint bundles written by the
 * Estimator (Training.py:944-1000 model_dir).  Returns the plain (unmasked) CRC through *out. */
int dd_crc32c(const void* data, size_t n, uint32_t crc, uint32_t* out);

/* ---- probes used by the test-suite to pin hardware fragment layouts the kernels rely on */
int dd_probe_tr16(const uint16_t* lds_image_4096, const int32_t* lane_byte_addr_64, uint16_t* out_64x4, dd_stream stream);

#ifdef __cplusplus
}
#endif
#endif /* DD_HIP_H */
