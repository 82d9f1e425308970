/*
 * dvc_hip.h -- C ABI of libdvc_hip.so: the MI355X (gfx950) kernels behind the
 * DMC inter-frame path (motion estimation -> flow warp -> motion compensation
 * -> contextual autoencoder -> entropy-model likelihoods / symbols) plus the
 * host range coder.
 *
 * The reference (lumingzzz/DeepVideoCodec) exposes no FFI: its seam is the
 * Python class models.DMC (dmc/models/__init__.py:1) and every operator below
 * is a stock ATen / compressai call inside it.  Each entry point cites the
 * reference lines whose arithmetic it replaces (paths relative to dmc/models/).
 *
 * Conventions
 *  - All tensors are fp32 (symbols / indexes are int32), NCHW, rows contiguous
 *    (element stride 1 along W, W along H).  Batch and channel strides are
 *    explicit (in elements) so a tensor may be a channel-slice view of a larger
 *    buffer.  The caller owns all memory; outputs are pre-allocated.
 *  - Every call is asynchronous on `stream` (a hipStream_t passed as void*;
 *    NULL = the null stream); no hidden synchronisation, no global state.
 *  - Return value: 0 on success, negative DVC_E* on error; dvc_last_error()
 *    returns a thread-local message for the last failing call on this thread.
 */
#ifndef DVC_HIP_H
#define DVC_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DVC_OK 0
#define DVC_EINVAL (-1)   /* bad argument (shape / pointer / unsupported config) */
#define DVC_EHIP (-2)     /* HIP runtime error (launch failed, no device ...) */
#define DVC_ENOSPC (-3)   /* output buffer too small (host coder) */
#define DVC_EFORMAT (-4)  /* malformed bitstream (host coder) */

#define DVC_MAX_SRC 4

/* A read-only NCHW fp32 operand (possibly one part of a channel concat). */
typedef struct DvcSrc {
  const float* ptr;      /* element (n=0, c=0, y=0, x=0) */
  int64_t batch_stride;  /* elements between consecutive n */
  int64_t chan_stride;   /* elements between consecutive c (>= H*W) */
  int32_t channels;
  int32_t _pad;
} DvcSrc;

/* A writable NCHW fp32 result. */
typedef struct DvcDst {
  float* ptr;
  int64_t batch_stride;
  int64_t chan_stride;
} DvcDst;

/*
 * 2-D convolution, kernel 1/3/7, stride 1/2, zero padding k/2, with the
 * element-wise neighbours of the reference graph folded in:
 *
 *   x   = concat(src[0..n_src-1], dim=C)               torch.cat call sites
 *   x   = in_act ? leaky_relu(x, in_slope) : x         ResBlock.first_layer (layers.py:72,76)
 *   t   = conv2d(x, weight[Cout,Cin,k,k], bias)        nn.Conv2d everywhere
 *   t   = out_act ? leaky_relu(t, out_slope) : t       slope 0 == ReLU
 *   t  += residual * (res_scale ? res_scale[n,c] : 1)  x + out (layers.py:81,114,142,168,295-297)
 *   t  += residual2                                    context + context_out (video_model.py:62-64)
 *   out = pixel_shuffle ? PixelShuffle(2)(t) : t       subpel_conv* (layers.py:45-56)
 *
 * With pixel_shuffle, `out`, `residual` and `res_scale` are laid out for the
 * shuffled tensor (Cout/4 channels, 2H x 2W).
 */
/* Operand arithmetic of dvc_conv2d.  FP32: exact fp32 products on v_mfma_f32_32x32x2_f32 (the parity path).
 * BF16: operands rounded to bf16.  BF16X2 / FP16X2: each operand split into hi + lo 16-bit parts (16 / 22
 * significant bits), a product = hi*hi + hi*lo + lo*hi on the 16-bit matrix pipe.  FP16X2 operands saturate
 * beyond +-131008. */
enum { DVC_PRECISION_FP32 = 0, DVC_PRECISION_BF16 = 1, DVC_PRECISION_BF16X2 = 2, DVC_PRECISION_FP16X2 = 3 };

typedef struct DvcConv2d {
  DvcSrc src[DVC_MAX_SRC];
  int32_t n_src;
  int32_t batch, in_h, in_w;       /* input spatial size (all sources) */
  const float* weight_packed;      /* dvc_conv2d_pack_weight() image of the [Cout][Cin][k][k] weight,
                                      Cin = sum(src.channels); 16-byte aligned */
  const float* bias;               /* [Cout] or NULL */
  int32_t out_channels;            /* Cout of the convolution (before shuffle) */
  int32_t ksize;                   /* 1, 3 or 7 */
  int32_t stride;                  /* 1 or 2 */
  int32_t in_act, out_act;         /* 0 / 1 */
  float in_slope, out_slope;
  int32_t pixel_shuffle;           /* 0 or 1 (factor 2) */
  DvcDst out;
  const float* residual;           /* NULL or tensor in output layout */
  int64_t res_batch_stride, res_chan_stride;
  const float* res_scale;          /* NULL or [batch][out channels after shuffle] */
  const float* residual2;          /* NULL or a second tensor in output layout (needs residual) */
  int64_t res2_batch_stride, res2_chan_stride;
  int32_t precision;               /* DVC_PRECISION_*: how the products are formed (fp32 accumulation and fp32
                                      tensors in every mode); weight_packed must have been packed for the
                                      same mode.  Modes other than FP32: ksize 7 with stride 1 only */
  int32_t _pad;
  float* plane_sums;               /* optional [batch][Cout][dvc_conv2d_plane_sum_slots()]: with its stores, every
                                      workgroup also writes the sum of the values it stored per output channel --
                                      the first stage of the plane means the squeeze-excite gate of
                                      dmc/models/layers.py:267-280 needs (dvc_se_gate_from_sums finishes it), so
                                      the tensor is not read back for them.  Only where the query below is > 0 */
} DvcConv2d;

int dvc_conv2d(const DvcConv2d* p, void* stream);
/* Partial sums per output plane dvc_conv2d(p) writes to p->plane_sums (the value of that field is ignored here):
 * > 0 for 3x3 stride-1 fp32 launches made of whole tiles, with 16-byte rows and without residual / PixelShuffle;
 * 0 where the launch cannot emit them (use dvc_se_gate); negative on invalid parameters. */
int32_t dvc_conv2d_plane_sum_slots(const DvcConv2d* p);

/* Weight layout of the convolution kernel: per block of 64 (Cout > 32) or 32 output
 * channels and per K-chunk of input channels, one contiguous zero-padded slab
 * [(ci - c0) * k*k + tap][co - cb0], so a slab is fetched global -> LDS in 1 KiB pieces.
 * Re-pack whenever the nn.Conv2d weight (OIHW) changes.  The packed size depends on
 * (Cout, Cin, k, stride); dvc_conv2d_packed_weight_floats returns it (negative on error).
 * For 7x7 stride-1 layers with exactly 16 output channels a second image follows the first inside the
 * same buffer (the filters paired with a copy moved one tap row down, 8 x 7 taps x 32 columns per
 * input channel): dvc_conv2d uses it to compute two output rows per matrix-instruction pass where
 * the launch allows.  The returned size covers both. */
int64_t dvc_conv2d_packed_weight_floats(int32_t cout, int32_t cin, int32_t ksize, int32_t stride);
int dvc_conv2d_pack_weight(const float* weight_oihw, int32_t cout, int32_t cin, int32_t ksize,
                           int32_t stride, float* packed, void* stream);
/* Image of the weight of the layer's DATA-GRADIENT convolution (training, dmc/train.py:329: loss.backward() through
 * nn.Conv2d): the gradient w.r.t. the input of a layer with weight [cout][cin][k][k] is the stride-1 convolution of
 * the (zero-upsampled, for stride 2) output gradient with W'[ci][co][ky][kx] = W[co][ci][k-1-ky][k-1-kx].  Packs W'
 * straight from the layer's OIHW tensor; size = dvc_conv2d_packed_weight_floats(cin, cout, ksize, 1). */
int dvc_conv2d_pack_weight_dgrad(const float* weight_oihw, int32_t cout, int32_t cin, int32_t ksize,
                                 float* packed, void* stream);

/* The same for the reduced-operand modes (precision = DVC_PRECISION_BF16 / BF16X2 / FP16X2), one part (rounded
 * operands) or two (hi + lo split), per 16 input channels; the size is returned in 4-byte units.
 * 1x1 and stride-2 layers: one slab per chunk, [part][tap][8-channel octet 0/1][co][8 x 16-bit].
 * 3x3 / 7x7 stride-1 layers: one slab per (chunk, tap row), [part][kx][octet][co][8 x 16-bit] padded to whole
 * 4 KiB -- the kernel walks K in (chunk, tap row) steps and DMAs a step's slab into a small LDS ring ahead of
 * its use, so a 7x7 chunk's weights (200 KiB in two parts) are never resident at once. */
int64_t dvc_conv2d_reduced_packed_weight_floats(int32_t cout, int32_t cin, int32_t ksize, int32_t stride,
                                                int32_t precision);
int dvc_conv2d_reduced_pack_weight(const float* weight_oihw, int32_t cout, int32_t cin, int32_t ksize,
                                   int32_t stride, int32_t precision, float* packed, void* stream);

/* Backward bilinear warp, border clamp, flow in pixels (ch0 = x, ch1 = y):
 * layers.py:175-198 (`torch_warp` = grid_sample(bilinear, border, align_corners=True)). */
int dvc_flow_warp(const DvcSrc* src, const DvcSrc* flow, const DvcDst* out,
                  int32_t batch, int32_t h, int32_t w, void* stream);

/* out = scale * mean of each 2x2 block.  avg_pool2d(2,2) (layers.py:250-251) with
 * scale 1; bilineardownsacling(x)/2 (layers.py:201-206, video_model.py:499-500) with scale .5 */
int dvc_avg_pool2(const DvcSrc* src, const DvcDst* out, int32_t batch, int32_t h, int32_t w,
                  float scale, void* stream);

/* MaxPool2d(2,2): layers.py:303,325,328. */
int dvc_max_pool2(const DvcSrc* src, const DvcDst* out, int32_t batch, int32_t h, int32_t w,
                  void* stream);

/* out = scale * bilinear x2 upsample (align_corners=False): layers.py:209-214,257. */
int dvc_bilinear_up2(const DvcSrc* src, const DvcDst* out, int32_t batch, int32_t h, int32_t w,
                     float scale, void* stream);

/* One SpyNet level's input (layers.py:257-262), fused:
 *   flow_up = 2 * up2(flow_prev)   (flow_prev NULL => zeros)
 *   out[:,0:3] = im1, out[:,3:6] = warp(im2, flow_up), out[:,6:8] = flow_up
 * h, w are the level's (output) size; flow_prev is (h/2, w/2). */
int dvc_spynet_level_input(const DvcSrc* im1, const DvcSrc* im2, const DvcSrc* flow_prev,
                           const DvcDst* out, int32_t batch, int32_t h, int32_t w, void* stream);

/* Squeeze-excite gate (layers.py:267-280): gate[n,c] = sigmoid(W2 relu(W1 mean_hw(x[n,c]))).
 * fc1 [C/r][C], fc2 [C][C/r].  `partial` is caller scratch of >= dvc_se_scratch_floats() floats.
 * Deterministic (fixed-order two-stage reduction, no float atomics). */
int64_t dvc_se_scratch_floats(int32_t batch, int32_t channels);
int dvc_se_gate(const DvcSrc* x, int32_t batch, int32_t h, int32_t w, const float* fc1,
                const float* fc2, int32_t hidden, float* partial, float* gate, void* stream);
/* The same gate from plane sums a convolution wrote with its stores (DvcConv2d::plane_sums, `slots` values per
 * (n, c) plane of h x w pixels): two small launches instead of a pass over the tensor.  Deterministic; the
 * summation order differs from dvc_se_gate's, so the two agree to rounding, not bit for bit. */
int dvc_se_gate_from_sums(const float* sums, int32_t slots, int32_t batch, int32_t channels, int32_t h, int32_t w,
                          const float* fc1, const float* fc2, int32_t hidden, float* partial, float* gate,
                          void* stream);

/*
 * dvc_se_gate for callers that cannot allocate device memory (code under HIP graph capture, tight
 * serving loops, harnesses that forbid hipMalloc).  No counterpart in the reference.  Same arithmetic and
 * the same kernels as dvc_se_gate, but the partial plane sums go to a device array built into the library
 * (1 MiB per GPU, held until the process ends) instead of caller scratch, and the call itself makes no
 * hipMalloc / hipFree.  Limit: batch * channels <= DVC_SE_STATIC_MAX_PLANES, larger inputs are rejected with
 * DVC_EINVAL.  All calls on one GPU share that array: calls that overlap in time on the same GPU (different
 * streams) race; calls on one stream are ordered.  The first call on a device looks the array's address up
 * (hipGetSymbolAddress); make that first call outside of a graph capture.
 */
#define DVC_SE_STATIC_MAX_PLANES 4096
#define DVC_MAX_DEVICES 64
int dvc_se_gate_static(const DvcSrc* x, int32_t batch, int32_t h, int32_t w, const float* fc1,
                       const float* fc2, int32_t hidden, float* gate, void* stream);

/* Plain strided copy / elementwise helpers. */
int dvc_copy(const DvcSrc* src, const DvcDst* out, int32_t batch, int32_t h, int32_t w, void* stream);
int dvc_clamp(const DvcSrc* src, const DvcDst* out, int32_t batch, int32_t h, int32_t w,
              float lo, float hi, void* stream);

/*
 * Factorized prior (compressai EntropyBottleneck as used at video_model.py:220-224,
 * 392-396; parameters base_model.py:63): per-channel cumulative 1-3-3-3-3-1.
 *   z_hat   = round(z - median[c]) + median[c]      (mode 0)   or   z + noise (mode 1, noise given)
 *   lik     = max(|sigmoid(s*upper) - sigmoid(s*lower)|, 1e-9), logits at z_hat -/+ .5
 *   symbols = int32 round(z - median[c])            (optional, may be NULL)
 * matrices/biases/factors: the raw parameters packed per channel as
 * [m0(3) b0(3) f0(3) | m1(9) b1(3) f1(3) | m2 .. | m3 .. | m4(3) b4(1)] = 58 floats. */
#define DVC_EB_PARAMS_PER_CHANNEL 58
int dvc_eb_forward(const DvcSrc* z, const float* noise, const float* packed_params,
                   const float* medians, const DvcDst* z_hat, const DvcDst* likelihood,
                   int32_t* symbols, int32_t batch, int32_t h, int32_t w, void* stream);
/* z_hat = symbols + median[c] (EntropyBottleneck.decompress' dequantise, video_model.py:257). */
int dvc_eb_dequantize(const int32_t* symbols, const float* medians, const DvcDst* z_hat,
                      int32_t batch, int32_t channels, int32_t h, int32_t w, void* stream);

/* Checkerboard dual prior, first pass (video_model.py:184-189 / 356-361):
 * yhat1[:, :C/2] = (round(y0 - mu0) + mu0) on even sites, 0 elsewhere;
 * yhat1[:, C/2:] = same for the second half on odd sites. */
int dvc_dual_prior_pass1(const DvcSrc* y, const DvcSrc* means, const DvcSrc* scales,
                         const DvcDst* yhat1, int32_t batch, int32_t h, int32_t w, void* stream);

/* process_with_mask as the reference states it (video_model.py:161-167 / 333-339), for callers that use that
 * helper directly (the coding path uses the fused passes above / below): `mask` is one (h, w) plane of floats
 * shared by every sample and channel; means_hat = means * mask, scales_hat = scales * mask,
 * y_quant = round((y - means_hat) * mask), y_hat = y_quant + means_hat.  Any destination may have ptr NULL. */
int dvc_process_with_mask(const DvcSrc* y, const DvcSrc* means, const DvcSrc* scales, const float* mask,
                          const DvcDst* y_quant, const DvcDst* y_hat, const DvcDst* means_hat,
                          const DvcDst* scales_hat, int32_t batch, int32_t h, int32_t w, void* stream);

/* Second pass + merge + Gaussian likelihood (video_model.py:192-216,232 and compressai
 * GaussianConditional): `spatial` is the y_spatial_prior output (2C channels:
 * mu0, sigma0, mu1, sigma1).  Outputs (any may have ptr NULL): y_hat, means_hat,
 * scales_hat, likelihood (all C channels); noise (C ch, may be NULL) switches the
 * likelihood input to y + noise (training-mode rate, video_model.py:232).
 * compress-mode outputs (may be NULL): symbols[2][N][C/2][h][w], indexes same shape
 * (video_model.py:209-214,248-249), scale_table[n_scales] device pointer. */
int dvc_dual_prior_pass2(const DvcSrc* y, const DvcSrc* means, const DvcSrc* scales,
                         const DvcSrc* spatial, const float* noise, const DvcDst* y_hat,
                         const DvcDst* means_hat, const DvcDst* scales_hat,
                         const DvcDst* likelihood, int32_t* symbols, int32_t* indexes,
                         const float* scale_table, int32_t n_scales, int32_t batch, int32_t h,
                         int32_t w, void* stream);

/* Decoder side (video_model.py:268-289 / 443-464).
 * step 0: indexes of plane 0 from hyper (means, scales).
 * step 1: yhat1 from decoded plane-0 symbols.
 * step 2: indexes of plane 1 from the spatial prior output.
 * step 3: y_hat from yhat1 + decoded plane-1 symbols. */
int dvc_dual_prior_dec_indexes(const DvcSrc* scales_or_spatial, int32_t plane, int32_t* indexes,
                               const float* scale_table, int32_t n_scales, int32_t batch,
                               int32_t channels, int32_t h, int32_t w, void* stream);
int dvc_dual_prior_dec_apply(const int32_t* symbols, const DvcSrc* means_or_spatial, int32_t plane,
                             const DvcSrc* yhat1_in, const DvcDst* out, int32_t batch,
                             int32_t channels, int32_t h, int32_t w, void* stream);

/* Stand-alone Gaussian conditional (compressai GaussianConditional.forward as called at
 * video_model.py:232,405): outputs = round(x - mu) + mu (or x + noise when noise != NULL,
 * noise contiguous like x); lik = max(Phi((.5-|o-mu|)/s) - Phi((-.5-|o-mu|)/s), 1e-9), s = max(scale,.11).
 * means may be NULL (zero). */
int dvc_gaussian_conditional(const DvcSrc* x, const DvcSrc* scales, const DvcSrc* means,
                             const float* noise, const DvcDst* outputs, const DvcDst* likelihood,
                             int32_t batch, int32_t h, int32_t w, void* stream);

/* GaussianConditional.build_indexes (video_model.py:248-249): index of the scale-table bin,
 * idx = (n_scales-1) - #{k < n_scales-1 : max(scale,.11) <= table[k]}; indexes contiguous like scales. */
int dvc_build_indexes(const DvcSrc* scales, int32_t* indexes, const float* scale_table,
                      int32_t n_scales, int32_t batch, int32_t h, int32_t w, void* stream);

/* bits[n] = sum over (c,h,w) of -log2(lik[n,...])  (dmc/train.py:83).  Deterministic
 * two-stage wavefront-shuffle reduction; `partial` >= dvc_bits_scratch_floats() floats. */
int64_t dvc_bits_scratch_floats(int32_t batch);
int dvc_likelihood_bits(const DvcSrc* lik, int32_t batch, int32_t h, int32_t w, float* partial,
                        float* bits, void* stream);

/* sse[n] = sum (a-b)^2 over (c,h,w) (PSNR numerator, dmc/test.py:205). Same scratch rule. */
int dvc_sq_err_sum(const DvcSrc* a, const DvcSrc* b, int32_t batch, int32_t h, int32_t w,
                   float* partial, float* sse, void* stream);

/*
 * ---- Gradients (training path).  The reference gets all of these from torch.autograd
 * (loss.backward() at dmc/train.py:329-336); each entry point is the adjoint of one forward
 * entry point above and cites the forward op's reference lines.
 */

/*
 * Weight gradient of dvc_conv2d (nn.Conv2d backward w.r.t. weight):
 *   gW[co][ci][ky][kx] = sum_{n,oy,ox} grad_out[n][co][oy][ox] * act_in(cat(srcs))[n][ci][oy*s+ky-p][ox*s+kx-p]
 * grad_out is the gradient before the output activation, contiguous (N, Cout, ho, wo) -- see
 * dvc_act_grad.  scratch needs dvc_conv2d_wgrad_scratch_floats() floats; grad_weight is OIHW.
 * grad_bias (Cout floats, or NULL to skip) receives the bias gradient sum_{n,oy,ox} grad_out[n][co][oy][ox]
 * from the same pass over grad_out.  Deterministic (fixed summation order).
 */
int64_t dvc_conv2d_wgrad_scratch_floats(int32_t cout, int32_t cin, int32_t ksize, int32_t stride,
                                        int32_t batch, int32_t in_h, int32_t in_w);
int dvc_conv2d_wgrad(const DvcSrc* srcs, int32_t n_src, int32_t batch, int32_t in_h, int32_t in_w,
                     int32_t in_act, float in_slope, const float* grad_out, int32_t out_channels,
                     int32_t ksize, int32_t stride, float* scratch, float* grad_weight, float* grad_bias,
                     void* stream);
/* grad_pre (contiguous N, C, ho, wo) = grad * act'(act_out), act'(t) = t > 0 ? 1 : slope, with
 * grad / act_out in the stored layout (PixelShuffle(2)-ed when pixel_shuffle); act_out NULL = no
 * activation (pure un-shuffle / compaction).  LeakyReLU / ReLU / PixelShuffle backward. */
int dvc_act_grad(const DvcSrc* grad, const DvcSrc* act_out, float slope, int32_t pixel_shuffle,
                 float* grad_pre, int32_t batch, int32_t channels, int32_t ho, int32_t wo, void* stream);
/* out = grad * (x > 0 ? 1 : slope): derivative of the fused input activation. */
int dvc_mul_act_grad(const DvcSrc* grad, const DvcSrc* x, float slope, const DvcDst* out,
                     int32_t batch, int32_t h, int32_t w, void* stream);
/* out = grad * (x > 0 ? 1 : slope) + plus (plus NULL: the function above).  x of a ResBlock feeds its first
 * convolution through an activation and the block's sum directly (layers.py:75-81): its gradient is the masked
 * data gradient plus the block's output gradient, formed here in one pass instead of by an autograd accumulation. */
int dvc_mul_act_grad_add(const DvcSrc* grad, const DvcSrc* x, float slope, const DvcSrc* plus, const DvcDst* out,
                         int32_t batch, int32_t h, int32_t w, void* stream);
/* out (out_h x out_w) = src (h x w) on the even sites, zero elsewhere: turns the data gradient
 * of a stride-2 convolution into a stride-1 convolution with the flipped kernel. */
int dvc_zero_upsample2(const DvcSrc* src, const DvcDst* out, int32_t batch, int32_t h, int32_t w,
                       int32_t out_h, int32_t out_w, void* stream);
/* out = a + b * scale[n*C + c] (+ c): un-fused residual / squeeze-excite adds (layers.py:72-81,
 * 280, 297); scale and c may be NULL.  dvc_scale_planes: out = x * scale[n*C + c]. */
int dvc_add_scaled(const DvcSrc* a, const DvcSrc* b, const float* scale, const DvcSrc* c,
                   const DvcDst* out, int32_t batch, int32_t h, int32_t w, void* stream);
int dvc_scale_planes(const DvcSrc* x, const float* scale, const DvcDst* out, int32_t batch,
                     int32_t h, int32_t w, void* stream);
/* out[n*C + c] = sum_hw a*b (b NULL: sum_hw a); sum_batch != 0: out[c] = sum over n as well
 * (bias gradient).  Two-stage, fixed order.  scratch: dvc_plane_dot_scratch_floats() floats. */
int64_t dvc_plane_dot_scratch_floats(int32_t batch, int32_t channels);
int dvc_plane_dot(const DvcSrc* a, const DvcSrc* b, int32_t batch, int32_t h, int32_t w,
                  int32_t sum_batch, float* scratch, float* out, void* stream);
/* Adjoint of dvc_flow_warp (layers.py:175-198): grad_src (pre-zeroed by the caller, accumulated
 * with float atomics) and / or grad_flow; either destination may be NULL. */
int dvc_flow_warp_backward(const DvcSrc* src, const DvcSrc* flow, const DvcSrc* grad_out,
                           const DvcDst* grad_src, const DvcDst* grad_flow, int32_t batch,
                           int32_t h, int32_t w, void* stream);
/* Adjoints of dvc_avg_pool2 / dvc_max_pool2 / dvc_bilinear_up2; h, w are the INPUT dims of the
 * forward op (so grad_in is h x w). */
int dvc_avg_pool2_backward(const DvcSrc* grad_out, const DvcDst* grad_in, int32_t batch, int32_t h,
                           int32_t w, float scale, void* stream);
int dvc_max_pool2_backward(const DvcSrc* x, const DvcSrc* grad_out, const DvcDst* grad_in,
                           int32_t batch, int32_t h, int32_t w, void* stream);
int dvc_bilinear_up2_backward(const DvcSrc* grad_out, const DvcDst* grad_in, int32_t batch,
                              int32_t h, int32_t w, float scale, void* stream);

/*
 * Gradient of dvc_eb_forward's likelihood (compressai EntropyBottleneck._likelihood +
 * LowerBound(1e-9)): x_eval = the tensor the likelihood was evaluated at (dvc_eb_forward's z_hat
 * output: z + noise in training).  grad_x may be NULL.  grad_packed is (C, 58), the gradient
 * w.r.t. the RAW packed parameters (softplus / tanh derivatives applied).  Deterministic.
 */
int64_t dvc_eb_backward_scratch_floats(int32_t channels);
int dvc_eb_backward(const DvcSrc* x_eval, const float* packed_params, const DvcSrc* grad_lik,
                    const DvcDst* grad_x, float* scratch, float* grad_packed, int32_t batch,
                    int32_t h, int32_t w, void* stream);
/* Adjoints of dvc_dual_prior_pass1 / pass2 (video_model.py:152-216 under autograd; round() is
 * the straight-through identity of utils.py:149-152, so y_hat passes its gradient to y only).
 * pass2: grad_yhat / grad_lik may be NULL; all four destinations are written in full (zeros at
 * the sites a tensor does not feed): grad_means / grad_scales at first-pass sites, grad_spatial
 * (2C channels) at second-pass sites. */
int dvc_dual_prior_pass1_backward(const DvcSrc* grad_yhat1, const DvcDst* grad_y, int32_t batch,
                                  int32_t h, int32_t w, void* stream);
int dvc_dual_prior_pass2_backward(const DvcSrc* y, const DvcSrc* means, const DvcSrc* scales,
                                  const DvcSrc* spatial, const float* noise, const DvcSrc* grad_yhat,
                                  const DvcSrc* grad_lik, const DvcDst* grad_y, const DvcDst* grad_means,
                                  const DvcDst* grad_scales, const DvcDst* grad_spatial, int32_t batch,
                                  int32_t h, int32_t w, void* stream);

/* Adjoints of the loss reductions dvc_likelihood_bits / dvc_sq_err_sum (RateDistortionLoss,
 * dmc/train.py:74-93,114-143): grad_lik = -grad_bits[n] / (ln2 * lik); grad_a = 2 grad_sse[n] (a - b). */
int dvc_likelihood_bits_backward(const DvcSrc* lik, const float* grad_bits, const DvcDst* grad_lik,
                                 int32_t batch, int32_t h, int32_t w, void* stream);
int dvc_sq_err_backward(const DvcSrc* a, const DvcSrc* b, const float* grad_sse, const DvcDst* grad_a,
                        int32_t batch, int32_t h, int32_t w, void* stream);

/*
 * Host range coder (no GPU involved): restates compressai.ans
 * RansEncoder.encode_with_indexes / RansDecoder.decode_with_indexes as used at
 * video_model.py:238-239,250-251,257,273,283.  64-bit-state rANS, 32-bit words,
 * 16-bit CDF precision, 4-bit bypass escape.  cdfs is [n_cdfs][cdf_stride] int32.
 * dvc_rans_encode returns the number of bytes written (>= 8) or a negative DVC_E*.
 */
int64_t dvc_rans_encode(const int32_t* symbols, const int32_t* indexes, int64_t n,
                        const int32_t* cdfs, int32_t n_cdfs, int32_t cdf_stride,
                        const int32_t* cdf_sizes, const int32_t* offsets, uint8_t* out,
                        int64_t out_capacity);
int dvc_rans_decode(const uint8_t* data, int64_t n_bytes, const int32_t* indexes, int64_t n,
                    const int32_t* cdfs, int32_t n_cdfs, int32_t cdf_stride,
                    const int32_t* cdf_sizes, const int32_t* offsets, int32_t* symbols_out);
/* Integer CDF of total 2^precision with every frequency >= 1 (compressai _CXX
 * pmf_to_quantized_cdf).  cdf_out has n_pmf + 1 entries. */
int dvc_pmf_to_quantized_cdf(const float* pmf, int32_t n_pmf, int32_t precision, int32_t* cdf_out);

const char* dvc_last_error(void);
/* Library / device introspection: number of visible HIP devices (<0 on error). */
int dvc_device_count(void);
const char* dvc_version(void);

#ifdef __cplusplus
}
#endif
#endif /* DVC_HIP_H */
