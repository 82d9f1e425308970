// Entropy-model kernels of the DMC path: factorized prior (per-channel cumulative
// network), checkerboard dual-prior quantisation passes, Gaussian-conditional
// likelihood, scale-table index search, and the wavefront-shuffle bit-count /
// squared-error reductions.  All element-wise over the (N, C, h, w) latents
// (68 x 120 at 1080p): one lane per element, everything one pass fuses is done
// in registers so each latent is read once and each output written once.
#include "dvc_common.h"

namespace {

using dvc::stream_grid;

struct Src { const float* p; long long bs, cs; int c; };
struct Dst { float* p; long long bs, cs; };
inline Src mk(const DvcSrc* s) { return s ? Src{s->ptr, s->batch_stride, s->chan_stride, s->channels} : Src{nullptr, 0, 0, 0}; }
inline Dst mk(const DvcDst* d) { return d ? Dst{d->ptr, d->batch_stride, d->chan_stride} : Dst{nullptr, 0, 0}; }

constexpr float LIK_BOUND = 1e-9f;
constexpr float SCALE_BOUND = 0.11f;

__device__ __forceinline__ float softplusf(float x) { return x > 20.f ? x : log1pf(expf(x)); }
__device__ __forceinline__ float sigmoidf(float x) { return 1.0f / (1.0f + expf(-x)); }
// Phi(x) = .5 * erfc(-x / sqrt 2)
__device__ __forceinline__ float std_cdf(float x) { return 0.5f * erfcf(-0.70710678118654752440f * x); }

// Cumulative logits of the 1-3-3-3-3-1 network for one channel; P is the channel's
// 58 transformed parameters (softplus already applied to matrices, tanh to factors).
__device__ __forceinline__ float eb_logits(const float* P, float x) {
  float a[3], b[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    float t = P[j] * x + P[3 + j];
    a[j] = t + P[6 + j] * tanhf(t);
  }
  const float* Q = P + 9;
#pragma unroll
  for (int l = 0; l < 3; ++l) {
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      // matmul row j: sum_k m[j][k] * a[k]   (k ascending, like torch.matmul's small GEMM)
      float t = Q[3 * j] * a[0] + Q[3 * j + 1] * a[1] + Q[3 * j + 2] * a[2] + Q[9 + j];
      b[j] = t + Q[12 + j] * tanhf(t);
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) a[j] = b[j];
    Q += 15;
  }
  return Q[0] * a[0] + Q[1] * a[1] + Q[2] * a[2] + Q[3];
}

// raw [C][58] -> transformed [C][58] in LDS, for the channels this block touches
__device__ __forceinline__ float eb_transform(int k, float v) {
  // layout per channel: m0(3) b0(3) f0(3) | 3 x { m(9) b(3) f(3) } | m4(3) b4(1)
  bool is_m, is_f;
  if (k < 9) { is_m = k < 3; is_f = k >= 6; }
  else if (k < 54) { const int r = (k - 9) % 15; is_m = r < 9; is_f = r >= 12; }
  else { is_m = k < 57; is_f = false; }
  return is_m ? softplusf(v) : (is_f ? tanhf(v) : v);
}

// grid: (blocks over N*h*w, C)
__global__ __launch_bounds__(256) void eb_forward_kernel(Src z, const float* noise, const float* packed,
                                                         const float* medians, Dst z_hat, Dst lik, int* symbols,
                                                         int batch, int h, int w) {
  __shared__ float P[DVC_EB_PARAMS_PER_CHANNEL];
  const int c = blockIdx.y;
  if (threadIdx.x < DVC_EB_PARAMS_PER_CHANNEL)
    P[threadIdx.x] = eb_transform(threadIdx.x, packed[c * DVC_EB_PARAMS_PER_CHANNEL + threadIdx.x]);
  __syncthreads();
  const float med = medians[c];
  const long long hw = (long long)h * w, total = hw * batch;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int n = (int)(i / hw);
    const long long pix = i - n * hw;
    const float v = z.p[n * z.bs + c * z.cs + pix];
    const float q = rintf(v - med);
    float out;
    if (noise != nullptr) out = v + noise[((long long)n * z.c + c) * hw + pix];
    else out = q + med;
    if (symbols != nullptr) symbols[((long long)n * z.c + c) * hw + pix] = (int)q;
    if (z_hat.p != nullptr) z_hat.p[n * z_hat.bs + c * z_hat.cs + pix] = out;
    if (lik.p != nullptr) {
      const float lower = eb_logits(P, out - 0.5f);
      const float upper = eb_logits(P, out + 0.5f);
      const float t = lower + upper;
      const float sgn = t > 0.f ? -1.f : (t < 0.f ? 1.f : 0.f);
      float l = fabsf(sigmoidf(sgn * upper) - sigmoidf(sgn * lower));
      lik.p[n * lik.bs + c * lik.cs + pix] = fmaxf(l, LIK_BOUND);
    }
  }
}

__global__ __launch_bounds__(256) void eb_dequantize_kernel(const int* symbols, const float* medians, Dst z_hat,
                                                            int batch, int channels, int h, int w) {
  const long long hw = (long long)h * w, total = hw * batch * channels;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long pix = i % hw;
    long long t = i / hw;
    const int c = (int)(t % channels), n = (int)(t / channels);
    z_hat.p[n * z_hat.bs + c * z_hat.cs + pix] = (float)symbols[i] + medians[c];
  }
}

// index = (n_scales - 1) - #{ s in table[0 .. n_scales-2] : max(scale, .11) <= s }
__device__ __forceinline__ int scale_index(float scale, const float* table, int n_scales) {
  const float s = fmaxf(scale, SCALE_BOUND);
  int idx = n_scales - 1;
  for (int k = 0; k < n_scales - 1; ++k) idx -= (s <= table[k]) ? 1 : 0;
  return idx;
}

#define DVC_DECODE_NCHW(i, C, hw, w)            \
  const long long pix = (i) % (hw);             \
  long long t_ = (i) / (hw);                    \
  const int c = (int)(t_ % (C));                \
  const int n = (int)(t_ / (C));                \
  const int yy = (int)(pix / (w));              \
  const int xx = (int)(pix - (long long)yy * (w)); \
  const int par = (yy + xx) & 1

__global__ __launch_bounds__(256) void dual_prior_pass1_kernel(Src y, Src means, Src scales, Dst yhat1, int batch,
                                                               int h, int w) {
  const int C = y.c, half = C / 2;
  const long long hw = (long long)h * w, total = hw * C * batch;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    DVC_DECODE_NCHW(i, C, hw, w);
    const int hi = c >= half ? 1 : 0;
    float out = 0.f;
    if (par == hi) {  // first-pass site: half 0 on even sites, half 1 on odd sites
      const float mu = means.p[n * means.bs + c * means.cs + pix];
      const float v = y.p[n * y.bs + c * y.cs + pix];
      out = rintf(v - mu) + mu;
    }
    yhat1.p[n * yhat1.bs + c * yhat1.cs + pix] = out;
  }
}

__global__ __launch_bounds__(256) void dual_prior_pass2_kernel(Src y, Src means, Src scales, Src sp,
                                                               const float* noise, Dst y_hat, Dst means_hat,
                                                               Dst scales_hat, Dst lik, int* symbols, int* indexes,
                                                               const float* scale_table, int n_scales, int batch,
                                                               int h, int w) {
  __shared__ float tab[128];
  if (scale_table != nullptr && threadIdx.x < n_scales) tab[threadIdx.x] = scale_table[threadIdx.x];
  __syncthreads();
  const int C = y.c, half = C / 2;
  const long long hw = (long long)h * w, total = hw * C * batch;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    DVC_DECODE_NCHW(i, C, hw, w);
    const int hi = c >= half ? 1 : 0;
    const int cl = c - hi * half;
    float mu, sg;
    const bool first = (par == hi);
    if (first) {
      mu = means.p[n * means.bs + c * means.cs + pix];
      sg = scales.p[n * scales.bs + c * scales.cs + pix];
    } else {  // spatial prior chunks: mu0, sigma0, mu1, sigma1 (each C/2 channels)
      const long long base = n * sp.bs + pix;
      mu = sp.p[base + (long long)(2 * hi * half + cl) * sp.cs];
      sg = sp.p[base + (long long)(2 * hi * half + half + cl) * sp.cs];
    }
    const float v = y.p[n * y.bs + c * y.cs + pix];
    const float q = rintf(v - mu);
    const float yh = q + mu;
    if (y_hat.p) y_hat.p[n * y_hat.bs + c * y_hat.cs + pix] = yh;
    if (means_hat.p) means_hat.p[n * means_hat.bs + c * means_hat.cs + pix] = mu;
    if (scales_hat.p) scales_hat.p[n * scales_hat.bs + c * scales_hat.cs + pix] = sg;
    if (lik.p) {
      const float in = noise ? v + noise[((long long)n * C + c) * hw + pix] : yh;
      const float s = fmaxf(sg, SCALE_BOUND);
      const float a = fabsf(in - mu);
      const float l = std_cdf((0.5f - a) / s) - std_cdf((-0.5f - a) / s);
      lik.p[n * lik.bs + c * lik.cs + pix] = fmaxf(l, LIK_BOUND);
    }
    if (symbols) {
      // plane 0 holds first-pass sites, plane 1 second-pass sites; both (N, C/2, h, w)
      const long long o = (((long long)(first ? 0 : 1) * batch + n) * half + cl) * hw + pix;
      symbols[o] = (int)q;
      indexes[o] = scale_index(sg, tab, n_scales);
    }
  }
}

// decoder: indexes of one plane.  plane 0: scales of the hyper prior (C channels: half 0 on
// even sites, half 1 on odd sites); plane 1: spatial prior output (2C channels), the complement.
__global__ __launch_bounds__(256) void dec_indexes_kernel(Src s, int plane, int* indexes, const float* scale_table,
                                                          int n_scales, int batch, int C, int h, int w) {
  __shared__ float tab[128];
  if (threadIdx.x < n_scales) tab[threadIdx.x] = scale_table[threadIdx.x];
  __syncthreads();
  const int half = C / 2;
  const long long hw = (long long)h * w, total = hw * half * batch;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    DVC_DECODE_NCHW(i, half, hw, w);
    // plane 0: site parity p -> half p.  plane 1: site parity p -> half 1-p.
    const int hi = plane == 0 ? par : 1 - par;
    const long long ch = plane == 0 ? (long long)(hi * half + c) : (long long)(2 * hi * half + half + c);
    indexes[i] = scale_index(s.p[n * s.bs + ch * s.cs + pix], tab, n_scales);
  }
}

// decoder: plane 0: yhat1 = (q + mu) on first-pass sites, 0 elsewhere (C channels).
//          plane 1: y_hat  = yhat1 + (q + mu_spatial) on second-pass sites.
__global__ __launch_bounds__(256) void dec_apply_kernel(const int* symbols, Src m, int plane, Src yhat1, Dst out,
                                                        int batch, int C, int h, int w) {
  const int half = C / 2;
  const long long hw = (long long)h * w, total = hw * C * batch;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    DVC_DECODE_NCHW(i, C, hw, w);
    const int hi = c >= half ? 1 : 0;
    const int cl = c - hi * half;
    const bool first = (par == hi);
    float v;
    if (plane == 0) {
      v = 0.f;
      if (first) {
        const float q = (float)symbols[((long long)n * half + cl) * hw + pix];
        v = q + m.p[n * m.bs + c * m.cs + pix];
      }
    } else {
      if (first) {
        v = yhat1.p[n * yhat1.bs + c * yhat1.cs + pix];
      } else {
        const float q = (float)symbols[((long long)n * half + cl) * hw + pix];
        v = q + m.p[n * m.bs + (long long)(2 * hi * half + cl) * m.cs + pix];
      }
    }
    out.p[n * out.bs + c * out.cs + pix] = v;
  }
}

// process_with_mask of the reference (video_model.py:161-167) for an arbitrary (h, w) mask plane shared by every
// sample and channel: means_hat = mu * m, scales_hat = sigma * m, y_quant = round((y - means_hat) * m),
// y_hat = y_quant + means_hat
__global__ __launch_bounds__(256) void process_with_mask_kernel(Src y, Src means, Src scales, const float* mask, Dst y_quant,
                                                                Dst y_hat, Dst means_hat, Dst scales_hat, int batch, int h,
                                                                int w) {
  const int C = y.c;
  const long long hw = (long long)h * w, total = hw * C * batch;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long pix = i % hw;
    long long t = i / hw;
    const int c = (int)(t % C), n = (int)(t / C);
    const float m = mask[pix];
    const float mh = __fmul_rn(means.p[n * means.bs + c * means.cs + pix], m);
    const float sh = __fmul_rn(scales.p[n * scales.bs + c * scales.cs + pix], m);
    const float q = rintf(__fmul_rn(__fsub_rn(y.p[n * y.bs + c * y.cs + pix], mh), m));
    if (y_quant.p) y_quant.p[n * y_quant.bs + c * y_quant.cs + pix] = q;
    if (y_hat.p) y_hat.p[n * y_hat.bs + c * y_hat.cs + pix] = __fadd_rn(q, mh);
    if (means_hat.p) means_hat.p[n * means_hat.bs + c * means_hat.cs + pix] = mh;
    if (scales_hat.p) scales_hat.p[n * scales_hat.bs + c * scales_hat.cs + pix] = sh;
  }
}

// stand-alone Gaussian conditional (compressai GaussianConditional.forward): outputs and likelihood
__global__ __launch_bounds__(256) void gaussian_conditional_kernel(Src x, Src scales, Src means, const float* noise,
                                                                   Dst outputs, Dst lik, int batch, int h, int w) {
  const int C = x.c;
  const long long hw = (long long)h * w, total = hw * C * batch;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long pix = i % hw;
    long long t = i / hw;
    const int c = (int)(t % C), n = (int)(t / C);
    const float v = x.p[n * x.bs + c * x.cs + pix];
    const float mu = means.p ? means.p[n * means.bs + c * means.cs + pix] : 0.f;
    const float out = noise ? v + noise[i] : rintf(v - mu) + mu;
    if (outputs.p) outputs.p[n * outputs.bs + c * outputs.cs + pix] = out;
    if (lik.p) {
      const float s = fmaxf(scales.p[n * scales.bs + c * scales.cs + pix], SCALE_BOUND);
      const float a = fabsf(out - mu);
      const float l = std_cdf((0.5f - a) / s) - std_cdf((-0.5f - a) / s);
      lik.p[n * lik.bs + c * lik.cs + pix] = fmaxf(l, LIK_BOUND);
    }
  }
}

__global__ __launch_bounds__(256) void build_indexes_kernel(Src s, int* indexes, const float* scale_table,
                                                            int n_scales, int batch, int h, int w) {
  __shared__ float tab[128];
  if (threadIdx.x < n_scales) tab[threadIdx.x] = scale_table[threadIdx.x];
  __syncthreads();
  const int C = s.c;
  const long long hw = (long long)h * w, total = hw * C * batch;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long pix = i % hw;
    long long t = i / hw;
    const int c = (int)(t % C), n = (int)(t / C);
    indexes[i] = scale_index(s.p[n * s.bs + c * s.cs + pix], tab, n_scales);
  }
}

// ---------------------------------------------------------------- reductions
constexpr int RED_SLICES = 256;  // partial sums per batch item

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

__device__ __forceinline__ float block_sum_256(float v, float* lds4) {
  v = wave_sum(v);
  if ((threadIdx.x & 63) == 0) lds4[threadIdx.x >> 6] = v;
  __syncthreads();
  float r = 0.f;
  if (threadIdx.x == 0) r = (lds4[0] + lds4[1]) + (lds4[2] + lds4[3]);
  __syncthreads();
  return r;
}

template <int MODE>  // 0: -log2(a)   1: (a-b)^2
__global__ __launch_bounds__(256) void reduce_partial_kernel(Src a, Src b, int h, int w, float* partial) {
  __shared__ float red[4];
  const int slice = blockIdx.x, n = blockIdx.y;
  const long long hw = (long long)h * w, total = hw * a.c;
  const long long per = (total + RED_SLICES - 1) / RED_SLICES;
  const long long lo = slice * per, hi = lo + per < total ? lo + per : total;
  float s = 0.f;
  for (long long i = lo + threadIdx.x; i < hi; i += 256) {
    const long long pix = i % hw;
    const int c = (int)(i / hw);
    const float va = a.p[n * a.bs + c * a.cs + pix];
    if (MODE == 0) {
      s -= log2f(va);
    } else {
      const float d = va - b.p[n * b.bs + c * b.cs + pix];
      s += d * d;
    }
  }
  s = block_sum_256(s, red);
  if (threadIdx.x == 0) partial[(long long)n * RED_SLICES + slice] = s;
}

__global__ __launch_bounds__(256) void reduce_final_kernel(const float* partial, float* out) {
  __shared__ float red[4];
  const int n = blockIdx.x;
  float s = block_sum_256(partial[(long long)n * RED_SLICES + threadIdx.x], red);
  if (threadIdx.x == 0) out[n] = s;
}

int check_plane(const DvcSrc* s, int h, int w, const char* who) {
  DVC_REQUIRE(s != nullptr && s->ptr != nullptr && s->channels >= 1, "%s: null/empty source", who);
  DVC_REQUIRE(s->chan_stride >= (int64_t)h * w, "%s: channel stride < H*W", who);
  return DVC_OK;
}
int check_dst_opt(const DvcDst* d, int h, int w, const char* who) {
  if (d == nullptr || d->ptr == nullptr) return DVC_OK;
  DVC_REQUIRE(d->chan_stride >= (int64_t)h * w, "%s: destination channel stride < H*W", who);
  return DVC_OK;
}

}  // namespace

#define DVC_TRY(expr) do { int rc_ = (expr); if (rc_) return rc_; } while (0)

extern "C" int dvc_eb_forward(const DvcSrc* z, const float* noise, const float* packed_params, const float* medians,
                              const DvcDst* z_hat, const DvcDst* likelihood, int32_t* symbols, int32_t batch,
                              int32_t h, int32_t w, void* stream) {
  DVC_REQUIRE(batch >= 1 && h >= 1 && w >= 1, "dvc_eb_forward: bad dims");
  DVC_TRY(check_plane(z, h, w, "dvc_eb_forward(z)"));
  DVC_REQUIRE(packed_params && medians, "dvc_eb_forward: null parameters");
  DVC_REQUIRE(z->channels <= 65535, "dvc_eb_forward: too many channels");
  DVC_TRY(check_dst_opt(z_hat, h, w, "dvc_eb_forward(z_hat)"));
  DVC_TRY(check_dst_opt(likelihood, h, w, "dvc_eb_forward(likelihood)"));
  const long long per_c = (long long)batch * h * w;
  dim3 grid(stream_grid(per_c, 256), z->channels);
  hipLaunchKernelGGL(eb_forward_kernel, grid, dim3(256), 0, (hipStream_t)stream, mk(z), noise, packed_params, medians,
                     mk(z_hat), mk(likelihood), symbols, batch, h, w);
  return dvc::check_launch("eb_forward_kernel");
}

extern "C" int dvc_eb_dequantize(const int32_t* symbols, const float* medians, const DvcDst* z_hat, int32_t batch,
                                 int32_t channels, int32_t h, int32_t w, void* stream) {
  DVC_REQUIRE(batch >= 1 && channels >= 1 && h >= 1 && w >= 1, "dvc_eb_dequantize: bad dims");
  DVC_REQUIRE(symbols && medians && z_hat && z_hat->ptr, "dvc_eb_dequantize: null pointer");
  DVC_TRY(check_dst_opt(z_hat, h, w, "dvc_eb_dequantize"));
  const long long total = (long long)batch * channels * h * w;
  hipLaunchKernelGGL(eb_dequantize_kernel, dim3(stream_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, symbols,
                     medians, mk(z_hat), batch, channels, h, w);
  return dvc::check_launch("eb_dequantize_kernel");
}

extern "C" int dvc_dual_prior_pass1(const DvcSrc* y, const DvcSrc* means, const DvcSrc* scales, const DvcDst* yhat1,
                                    int32_t batch, int32_t h, int32_t w, void* stream) {
  DVC_REQUIRE(batch >= 1 && h >= 1 && w >= 1, "dvc_dual_prior_pass1: bad dims");
  DVC_TRY(check_plane(y, h, w, "dvc_dual_prior_pass1(y)"));
  DVC_TRY(check_plane(means, h, w, "dvc_dual_prior_pass1(means)"));
  DVC_TRY(check_plane(scales, h, w, "dvc_dual_prior_pass1(scales)"));
  DVC_REQUIRE(y->channels % 2 == 0 && means->channels == y->channels && scales->channels == y->channels,
              "dvc_dual_prior_pass1: channel mismatch");
  DVC_REQUIRE(yhat1 && yhat1->ptr, "dvc_dual_prior_pass1: null output");
  DVC_TRY(check_dst_opt(yhat1, h, w, "dvc_dual_prior_pass1(out)"));
  const long long total = (long long)batch * y->channels * h * w;
  hipLaunchKernelGGL(dual_prior_pass1_kernel, dim3(stream_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, mk(y),
                     mk(means), mk(scales), mk(yhat1), batch, h, w);
  return dvc::check_launch("dual_prior_pass1_kernel");
}

extern "C" int dvc_process_with_mask(const DvcSrc* y, const DvcSrc* means, const DvcSrc* scales, const float* mask,
                                     const DvcDst* y_quant, const DvcDst* y_hat, const DvcDst* means_hat,
                                     const DvcDst* scales_hat, int32_t batch, int32_t h, int32_t w, void* stream) {
  DVC_REQUIRE(batch >= 1 && h >= 1 && w >= 1, "dvc_process_with_mask: bad dims");
  DVC_TRY(check_plane(y, h, w, "dvc_process_with_mask(y)"));
  DVC_TRY(check_plane(means, h, w, "dvc_process_with_mask(means)"));
  DVC_TRY(check_plane(scales, h, w, "dvc_process_with_mask(scales)"));
  DVC_REQUIRE(means->channels == y->channels && scales->channels == y->channels, "dvc_process_with_mask: channel mismatch");
  DVC_REQUIRE(mask != nullptr, "dvc_process_with_mask: null mask");
  DVC_TRY(check_dst_opt(y_quant, h, w, "dvc_process_with_mask(y_quant)"));
  DVC_TRY(check_dst_opt(y_hat, h, w, "dvc_process_with_mask(y_hat)"));
  DVC_TRY(check_dst_opt(means_hat, h, w, "dvc_process_with_mask(means_hat)"));
  DVC_TRY(check_dst_opt(scales_hat, h, w, "dvc_process_with_mask(scales_hat)"));
  const long long total = (long long)batch * y->channels * h * w;
  hipLaunchKernelGGL(process_with_mask_kernel, dim3(stream_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, mk(y),
                     mk(means), mk(scales), mask, mk(y_quant), mk(y_hat), mk(means_hat), mk(scales_hat), batch, h, w);
  return dvc::check_launch("process_with_mask_kernel");
}

extern "C" int dvc_dual_prior_pass2(const DvcSrc* y, const DvcSrc* means, const DvcSrc* scales, const DvcSrc* spatial,
                                    const float* noise, const DvcDst* y_hat, const DvcDst* means_hat,
                                    const DvcDst* scales_hat, const DvcDst* likelihood, int32_t* symbols,
                                    int32_t* indexes, const float* scale_table, int32_t n_scales, int32_t batch,
                                    int32_t h, int32_t w, void* stream) {
  DVC_REQUIRE(batch >= 1 && h >= 1 && w >= 1, "dvc_dual_prior_pass2: bad dims");
  DVC_TRY(check_plane(y, h, w, "dvc_dual_prior_pass2(y)"));
  DVC_TRY(check_plane(means, h, w, "dvc_dual_prior_pass2(means)"));
  DVC_TRY(check_plane(scales, h, w, "dvc_dual_prior_pass2(scales)"));
  DVC_TRY(check_plane(spatial, h, w, "dvc_dual_prior_pass2(spatial)"));
  DVC_REQUIRE(y->channels % 2 == 0 && means->channels == y->channels && scales->channels == y->channels &&
                  spatial->channels == 2 * y->channels, "dvc_dual_prior_pass2: channel mismatch");
  DVC_REQUIRE((symbols == nullptr) == (indexes == nullptr), "dvc_dual_prior_pass2: symbols and indexes go together");
  DVC_REQUIRE(symbols == nullptr || (scale_table != nullptr && n_scales >= 2 && n_scales <= 128),
              "dvc_dual_prior_pass2: scale table (2..128 entries) required for indexes");
  DVC_TRY(check_dst_opt(y_hat, h, w, "dvc_dual_prior_pass2(y_hat)"));
  DVC_TRY(check_dst_opt(means_hat, h, w, "dvc_dual_prior_pass2(means_hat)"));
  DVC_TRY(check_dst_opt(scales_hat, h, w, "dvc_dual_prior_pass2(scales_hat)"));
  DVC_TRY(check_dst_opt(likelihood, h, w, "dvc_dual_prior_pass2(likelihood)"));
  const long long total = (long long)batch * y->channels * h * w;
  hipLaunchKernelGGL(dual_prior_pass2_kernel, dim3(stream_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, mk(y),
                     mk(means), mk(scales), mk(spatial), noise, mk(y_hat), mk(means_hat), mk(scales_hat),
                     mk(likelihood), symbols, indexes, symbols ? scale_table : nullptr, symbols ? n_scales : 0, batch,
                     h, w);
  return dvc::check_launch("dual_prior_pass2_kernel");
}

extern "C" int dvc_dual_prior_dec_indexes(const DvcSrc* s, int32_t plane, int32_t* indexes, const float* scale_table,
                                          int32_t n_scales, int32_t batch, int32_t channels, int32_t h, int32_t w,
                                          void* stream) {
  DVC_REQUIRE(batch >= 1 && h >= 1 && w >= 1 && channels >= 2 && channels % 2 == 0, "dvc_dual_prior_dec_indexes: bad dims");
  DVC_REQUIRE(plane == 0 || plane == 1, "dvc_dual_prior_dec_indexes: plane must be 0 or 1");
  DVC_TRY(check_plane(s, h, w, "dvc_dual_prior_dec_indexes"));
  DVC_REQUIRE(s->channels == (plane == 0 ? channels : 2 * channels), "dvc_dual_prior_dec_indexes: channel mismatch");
  DVC_REQUIRE(indexes && scale_table && n_scales >= 2 && n_scales <= 128, "dvc_dual_prior_dec_indexes: bad table");
  const long long total = (long long)batch * (channels / 2) * h * w;
  hipLaunchKernelGGL(dec_indexes_kernel, dim3(stream_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, mk(s), plane,
                     indexes, scale_table, n_scales, batch, channels, h, w);
  return dvc::check_launch("dec_indexes_kernel");
}

extern "C" int dvc_dual_prior_dec_apply(const int32_t* symbols, const DvcSrc* m, int32_t plane, const DvcSrc* yhat1_in,
                                        const DvcDst* out, int32_t batch, int32_t channels, int32_t h, int32_t w,
                                        void* stream) {
  DVC_REQUIRE(batch >= 1 && h >= 1 && w >= 1 && channels >= 2 && channels % 2 == 0, "dvc_dual_prior_dec_apply: bad dims");
  DVC_REQUIRE(plane == 0 || plane == 1, "dvc_dual_prior_dec_apply: plane must be 0 or 1");
  DVC_TRY(check_plane(m, h, w, "dvc_dual_prior_dec_apply(means)"));
  DVC_REQUIRE(m->channels == (plane == 0 ? channels : 2 * channels), "dvc_dual_prior_dec_apply: channel mismatch");
  if (plane == 1) {
    DVC_TRY(check_plane(yhat1_in, h, w, "dvc_dual_prior_dec_apply(yhat1)"));
    DVC_REQUIRE(yhat1_in->channels == channels, "dvc_dual_prior_dec_apply: yhat1 channel mismatch");
  }
  DVC_REQUIRE(symbols && out && out->ptr, "dvc_dual_prior_dec_apply: null pointer");
  DVC_TRY(check_dst_opt(out, h, w, "dvc_dual_prior_dec_apply(out)"));
  const long long total = (long long)batch * channels * h * w;
  hipLaunchKernelGGL(dec_apply_kernel, dim3(stream_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, symbols, mk(m),
                     plane, mk(plane == 1 ? yhat1_in : nullptr), mk(out), batch, channels, h, w);
  return dvc::check_launch("dec_apply_kernel");
}

extern "C" int dvc_gaussian_conditional(const DvcSrc* x, const DvcSrc* scales, const DvcSrc* means, const float* noise,
                                        const DvcDst* outputs, const DvcDst* likelihood, int32_t batch, int32_t h,
                                        int32_t w, void* stream) {
  DVC_REQUIRE(batch >= 1 && h >= 1 && w >= 1, "dvc_gaussian_conditional: bad dims");
  DVC_TRY(check_plane(x, h, w, "dvc_gaussian_conditional(x)"));
  DVC_TRY(check_plane(scales, h, w, "dvc_gaussian_conditional(scales)"));
  DVC_REQUIRE(scales->channels == x->channels, "dvc_gaussian_conditional: channel mismatch");
  if (means != nullptr && means->ptr != nullptr) {
    DVC_TRY(check_plane(means, h, w, "dvc_gaussian_conditional(means)"));
    DVC_REQUIRE(means->channels == x->channels, "dvc_gaussian_conditional: channel mismatch");
  }
  DVC_TRY(check_dst_opt(outputs, h, w, "dvc_gaussian_conditional(outputs)"));
  DVC_TRY(check_dst_opt(likelihood, h, w, "dvc_gaussian_conditional(likelihood)"));
  const long long total = (long long)batch * x->channels * h * w;
  hipLaunchKernelGGL(gaussian_conditional_kernel, dim3(stream_grid(total, 256)), dim3(256), 0, (hipStream_t)stream,
                     mk(x), mk(scales), mk(means), noise, mk(outputs), mk(likelihood), batch, h, w);
  return dvc::check_launch("gaussian_conditional_kernel");
}

extern "C" int dvc_build_indexes(const DvcSrc* scales, int32_t* indexes, const float* scale_table, int32_t n_scales,
                                 int32_t batch, int32_t h, int32_t w, void* stream) {
  DVC_REQUIRE(batch >= 1 && h >= 1 && w >= 1, "dvc_build_indexes: bad dims");
  DVC_TRY(check_plane(scales, h, w, "dvc_build_indexes"));
  DVC_REQUIRE(indexes && scale_table && n_scales >= 2 && n_scales <= 128, "dvc_build_indexes: bad table");
  const long long total = (long long)batch * scales->channels * h * w;
  hipLaunchKernelGGL(build_indexes_kernel, dim3(stream_grid(total, 256)), dim3(256), 0, (hipStream_t)stream,
                     mk(scales), indexes, scale_table, n_scales, batch, h, w);
  return dvc::check_launch("build_indexes_kernel");
}

extern "C" int64_t dvc_bits_scratch_floats(int32_t batch) { return (int64_t)batch * RED_SLICES; }

extern "C" int dvc_likelihood_bits(const DvcSrc* lik, int32_t batch, int32_t h, int32_t w, float* partial, float* bits,
                                   void* stream) {
  DVC_REQUIRE(batch >= 1 && batch <= 65535 && h >= 1 && w >= 1, "dvc_likelihood_bits: bad dims");
  DVC_TRY(check_plane(lik, h, w, "dvc_likelihood_bits"));
  DVC_REQUIRE(partial && bits, "dvc_likelihood_bits: null pointer");
  hipLaunchKernelGGL((reduce_partial_kernel<0>), dim3(RED_SLICES, batch), dim3(256), 0, (hipStream_t)stream, mk(lik),
                     mk(lik), h, w, partial);
  DVC_TRY(dvc::check_launch("reduce_partial_kernel<bits>"));
  hipLaunchKernelGGL(reduce_final_kernel, dim3(batch), dim3(256), 0, (hipStream_t)stream, (const float*)partial, bits);
  return dvc::check_launch("reduce_final_kernel");
}

extern "C" int dvc_sq_err_sum(const DvcSrc* a, const DvcSrc* b, int32_t batch, int32_t h, int32_t w, float* partial,
                              float* sse, void* stream) {
  DVC_REQUIRE(batch >= 1 && batch <= 65535 && h >= 1 && w >= 1, "dvc_sq_err_sum: bad dims");
  DVC_TRY(check_plane(a, h, w, "dvc_sq_err_sum(a)"));
  DVC_TRY(check_plane(b, h, w, "dvc_sq_err_sum(b)"));
  DVC_REQUIRE(a->channels == b->channels, "dvc_sq_err_sum: channel mismatch");
  DVC_REQUIRE(partial && sse, "dvc_sq_err_sum: null pointer");
  hipLaunchKernelGGL((reduce_partial_kernel<1>), dim3(RED_SLICES, batch), dim3(256), 0, (hipStream_t)stream, mk(a),
                     mk(b), h, w, partial);
  DVC_TRY(dvc::check_launch("reduce_partial_kernel<sse>"));
  hipLaunchKernelGGL(reduce_final_kernel, dim3(batch), dim3(256), 0, (hipStream_t)stream, (const float*)partial, sse);
  return dvc::check_launch("reduce_final_kernel");
}
