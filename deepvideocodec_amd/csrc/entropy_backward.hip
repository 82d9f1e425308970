// Gradients of the entropy-model kernels (training path): the factorized prior's likelihood
// w.r.t. its input and its 58 parameters per channel, and the checkerboard dual-prior pass
// (round-STE reconstruction + Gaussian likelihood) w.r.t. the latent, the hyper-prior
// means / scales and the spatial-prior output.  Gradient rules follow compressai's:
// LowerBound passes a gradient where the input is above the bound or the gradient pushes it up
// (grad < 0); round() is a straight-through identity (reference utils.py:149-152).
#include "dvc_common.h"

namespace {

using dvc::stream_grid;

struct Src { const float* p; long long bs, cs; int c; };
struct Dst { float* p; long long bs, cs; };
inline Src mk(const DvcSrc* s) { return s ? Src{s->ptr, s->batch_stride, s->chan_stride, s->channels} : Src{nullptr, 0, 0, 0}; }
inline Dst mk(const DvcDst* d) { return d ? Dst{d->ptr, d->batch_stride, d->chan_stride} : Dst{nullptr, 0, 0}; }

constexpr float LIK_BOUND = 1e-9f;
constexpr float SCALE_BOUND = 0.11f;
constexpr int NP = DVC_EB_PARAMS_PER_CHANNEL;   // 58
constexpr int EB_CHUNKS = 32;                   // partial parameter gradients per channel

__device__ __forceinline__ float softplusf(float x) { return x > 20.f ? x : log1pf(expf(x)); }
__device__ __forceinline__ float sigmoidf(float x) { return 1.0f / (1.0f + expf(-x)); }
__device__ __forceinline__ float std_cdf(float x) { return 0.5f * erfcf(-0.70710678118654752440f * x); }
__device__ __forceinline__ float std_pdf(float x) { return 0.39894228040143267794f * expf(-0.5f * x * x); }

// kind of raw parameter k of a channel: 0 = matrix (softplus), 1 = bias, 2 = factor (tanh)
__device__ __forceinline__ int eb_kind(int k) {
  if (k < 9) return k < 3 ? 0 : (k >= 6 ? 2 : 1);
  if (k < 54) { const int r = (k - 9) % 15; return r < 9 ? 0 : (r >= 12 ? 2 : 1); }
  return k < 57 ? 0 : 1;
}
__device__ __forceinline__ float eb_transform(int k, float v) {
  const int kind = eb_kind(k);
  return kind == 0 ? softplusf(v) : (kind == 2 ? tanhf(v) : v);
}
// d transform / d raw
__device__ __forceinline__ float eb_transform_grad(int k, float v) {
  const int kind = eb_kind(k);
  if (kind == 0) return sigmoidf(v);
  if (kind == 2) { const float t = tanhf(v); return 1.0f - t * t; }
  return 1.0f;
}

// Forward of the cumulative network at x keeping the activations, then the reverse sweep for an
// upstream gradient gl on the logits: adds the gradients w.r.t. the TRANSFORMED parameters into
// gP[58] and returns gl * dlogits/dx.  If gl is 0 only the logits are needed: use eb_logits.
struct EbTape { float a[4][3], th[4][3], logits; };

__device__ __forceinline__ void eb_forward_tape(const float* P, float x, EbTape& T) {
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const float t = P[j] * x + P[3 + j];
    T.th[0][j] = tanhf(t);
    T.a[0][j] = t + P[6 + j] * T.th[0][j];
  }
#pragma unroll
  for (int l = 1; l < 4; ++l) {
    const float* Q = P + 9 + 15 * (l - 1);
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const float t = Q[3 * j] * T.a[l - 1][0] + Q[3 * j + 1] * T.a[l - 1][1] + Q[3 * j + 2] * T.a[l - 1][2] + Q[9 + j];
      T.th[l][j] = tanhf(t);
      T.a[l][j] = t + Q[12 + j] * T.th[l][j];
    }
  }
  const float* Q4 = P + 54;
  T.logits = Q4[0] * T.a[3][0] + Q4[1] * T.a[3][1] + Q4[2] * T.a[3][2] + Q4[3];
}

__device__ __forceinline__ float eb_backward_tape(const float* P, float x, const EbTape& T, float gl, float* gP) {
  float ga[3], gprev[3];
  const float* Q4 = P + 54;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    gP[54 + k] += gl * T.a[3][k];
    ga[k] = gl * Q4[k];
  }
  gP[57] += gl;
#pragma unroll
  for (int l = 3; l >= 1; --l) {
    const float* Q = P + 9 + 15 * (l - 1);
    float* gQ = gP + 9 + 15 * (l - 1);
#pragma unroll
    for (int k = 0; k < 3; ++k) gprev[k] = 0.f;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      gQ[12 + j] += ga[j] * T.th[l][j];
      const float gt = ga[j] * (1.0f + Q[12 + j] * (1.0f - T.th[l][j] * T.th[l][j]));
      gQ[9 + j] += gt;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        gQ[3 * j + k] += gt * T.a[l - 1][k];
        gprev[k] += gt * Q[3 * j + k];
      }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) ga[k] = gprev[k];
  }
  float gx = 0.f;
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    gP[6 + j] += ga[j] * T.th[0][j];
    const float gt = ga[j] * (1.0f + P[6 + j] * (1.0f - T.th[0][j] * T.th[0][j]));
    gP[3 + j] += gt;
    gP[j] += gt * x;
    gx += gt * P[j];
  }
  return gx;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

// grid (EB_CHUNKS, C): block (chunk, c) walks the elements chunk, chunk + EB_CHUNKS*256 ... of
// channel c; parameter gradients are summed per block in a fixed order and written to
// partial[(c * EB_CHUNKS + chunk) * 58 + k].
__global__ __launch_bounds__(256) void eb_backward_kernel(Src x, const float* packed, Src glik, Dst gx, float* partial,
                                                          int batch, int h, int w) {
  __shared__ float P[NP];
  __shared__ float red[4][NP];
  const int c = blockIdx.y;
  if (threadIdx.x < NP) P[threadIdx.x] = eb_transform(threadIdx.x, packed[c * NP + threadIdx.x]);
  __syncthreads();
  float gP[NP];
#pragma unroll
  for (int k = 0; k < NP; ++k) gP[k] = 0.f;
  const long long hw = (long long)h * w, total = hw * batch;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int n = (int)(i / hw);
    const long long pix = i - n * hw;
    const float v = x.p[n * x.bs + c * x.cs + pix];
    float g = glik.p[n * glik.bs + c * glik.cs + pix];
    EbTape lo, up;
    eb_forward_tape(P, v - 0.5f, lo);
    eb_forward_tape(P, v + 0.5f, up);
    const float t = lo.logits + up.logits;
    const float sgn = t > 0.f ? -1.f : (t < 0.f ? 1.f : 0.f);
    const float su = sigmoidf(sgn * up.logits), sl = sigmoidf(sgn * lo.logits);
    const float d = su - sl;
    const float lik = fabsf(d);
    if (!(lik >= LIK_BOUND || g < 0.f)) g = 0.f;          // likelihood lower bound
    const float sd = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
    const float g_up = g * sd * sgn * su * (1.0f - su);
    const float g_lo = -g * sd * sgn * sl * (1.0f - sl);
    float gv = eb_backward_tape(P, v + 0.5f, up, g_up, gP);
    gv += eb_backward_tape(P, v - 0.5f, lo, g_lo, gP);
    if (gx.p != nullptr) gx.p[n * gx.bs + c * gx.cs + pix] = gv;
  }
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < NP; ++k) {
    const float s = wave_sum(gP[k]);
    if (lane == 0) red[wv][k] = s;
  }
  __syncthreads();
  if (threadIdx.x < NP)
    partial[((long long)c * EB_CHUNKS + blockIdx.x) * NP + threadIdx.x] =
        (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

__global__ __launch_bounds__(256) void eb_backward_final_kernel(const float* partial, const float* packed,
                                                                float* gpacked, int channels) {
  const int total = channels * NP;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
    const int c = i / NP, k = i - c * NP;
    float s = 0.f;
    for (int j = 0; j < EB_CHUNKS; ++j) s += partial[((long long)c * EB_CHUNKS + j) * NP + k];
    gpacked[i] = s * eb_transform_grad(k, packed[i]);
  }
}

#define DECODE_NCHW(i, C, hw, w)                \
  const long long pix = (i) % (hw);             \
  long long t_ = (i) / (hw);                    \
  const int c = (int)(t_ % (C));                \
  const int n = (int)(t_ / (C));                \
  const int yy = (int)(pix / (w));              \
  const int xx = (int)(pix - (long long)yy * (w)); \
  const int par = (yy + xx) & 1

// adjoint of the first checkerboard pass: y_hat1 = round-STE(y - mu) + mu on first-pass sites
__global__ __launch_bounds__(256) void dual_prior_pass1_backward_kernel(Src g, Dst gy, int batch, int h, int w) {
  const int C = g.c, half = C / 2;
  const long long hw = (long long)h * w, total = hw * C * batch;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    DECODE_NCHW(i, C, hw, w);
    const int hi = c >= half ? 1 : 0;
    gy.p[n * gy.bs + c * gy.cs + pix] = (par == hi) ? g.p[n * g.bs + c * g.cs + pix] : 0.f;
  }
}

// adjoint of the second pass (see dual_prior_pass2_kernel): y_hat = round-STE(y - mu) + mu at every
// site (mu, sigma from the hyper prior on first-pass sites, from the spatial prior elsewhere) and
// likelihood = Phi((.5 - |in - mu|) / s) - Phi((-.5 - |in - mu|) / s), s = max(sigma, .11),
// in = y + noise.
__global__ __launch_bounds__(256) void dual_prior_pass2_backward_kernel(Src y, Src means, Src scales, Src sp,
                                                                        const float* noise, Src g_yhat, Src g_lik,
                                                                        Dst gy, Dst gmeans, Dst gscales, Dst gsp,
                                                                        int batch, int h, int w) {
  const int C = y.c, half = C / 2;
  const long long hw = (long long)h * w, total = hw * C * batch;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    DECODE_NCHW(i, C, hw, w);
    const int hi = c >= half ? 1 : 0;
    const int cl = c - hi * half;
    const bool first = (par == hi);
    const long long sp_mu = n * sp.bs + pix + (long long)(2 * hi * half + cl) * sp.cs;
    const long long sp_sg = n * sp.bs + pix + (long long)(2 * hi * half + half + cl) * sp.cs;
    float mu, sg;
    if (first) {
      mu = means.p[n * means.bs + c * means.cs + pix];
      sg = scales.p[n * scales.bs + c * scales.cs + pix];
    } else {
      mu = sp.p[sp_mu];
      sg = sp.p[sp_sg];
    }
    const float v = y.p[n * y.bs + c * y.cs + pix];
    float g_in = 0.f, g_mu = 0.f, g_sg = 0.f;
    if (g_lik.p != nullptr) {
      float g = g_lik.p[n * g_lik.bs + c * g_lik.cs + pix];
      const float in = noise ? v + noise[((long long)n * C + c) * hw + pix] : rintf(v - mu) + mu;
      const float s = fmaxf(sg, SCALE_BOUND);
      const float dlt = in - mu;
      const float a = fabsf(dlt);
      const float u1 = (0.5f - a) / s, u2 = (-0.5f - a) / s;
      const float lik = std_cdf(u1) - std_cdf(u2);
      if (!(lik >= LIK_BOUND || g < 0.f)) g = 0.f;
      const float p1 = std_pdf(u1), p2 = std_pdf(u2);
      const float dl_da = (p2 - p1) / s;
      const float dl_ds = (u2 * p2 - u1 * p1) / s;
      const float sa = dlt > 0.f ? 1.f : (dlt < 0.f ? -1.f : 0.f);
      g_in = g * dl_da * sa;
      g_mu = -g_in;
      g_sg = g * dl_ds;
      if (!(sg >= SCALE_BOUND || g_sg < 0.f)) g_sg = 0.f;   // scale lower bound
    }
    float g_y = noise ? g_in : 0.f;   // without noise the likelihood sees round(): no gradient to y
    if (g_yhat.p != nullptr) g_y += g_yhat.p[n * g_yhat.bs + c * g_yhat.cs + pix];
    gy.p[n * gy.bs + c * gy.cs + pix] = g_y;
    gmeans.p[n * gmeans.bs + c * gmeans.cs + pix] = first ? g_mu : 0.f;
    gscales.p[n * gscales.bs + c * gscales.cs + pix] = first ? g_sg : 0.f;
    gsp.p[n * gsp.bs + pix + (long long)(2 * hi * half + cl) * gsp.cs] = first ? 0.f : g_mu;
    gsp.p[n * gsp.bs + pix + (long long)(2 * hi * half + half + cl) * gsp.cs] = first ? 0.f : g_sg;
  }
}

int check_plane(const DvcSrc* s, int h, int w, const char* who) {
  DVC_REQUIRE(s != nullptr && s->ptr != nullptr && s->channels >= 1, "%s: null/empty source", who);
  DVC_REQUIRE(s->chan_stride >= (int64_t)h * w, "%s: channel stride < H*W", who);
  return DVC_OK;
}
int check_dst(const DvcDst* d, int h, int w, const char* who) {
  DVC_REQUIRE(d != nullptr && d->ptr != nullptr && d->chan_stride >= (int64_t)h * w, "%s: bad destination", who);
  return DVC_OK;
}
#define CHECK(expr) do { int rc_ = (expr); if (rc_) return rc_; } while (0)

}  // namespace

extern "C" int64_t dvc_eb_backward_scratch_floats(int32_t channels) { return (int64_t)channels * EB_CHUNKS * NP; }

extern "C" int dvc_eb_backward(const DvcSrc* x_eval, const float* packed_params, const DvcSrc* grad_lik,
                               const DvcDst* grad_x, float* scratch, float* grad_packed, int32_t batch, int32_t h,
                               int32_t w, void* stream) {
  DVC_REQUIRE(batch >= 1 && h >= 1 && w >= 1, "dvc_eb_backward: bad dims");
  CHECK(check_plane(x_eval, h, w, "dvc_eb_backward(x)"));
  CHECK(check_plane(grad_lik, h, w, "dvc_eb_backward(grad_lik)"));
  DVC_REQUIRE(grad_lik->channels == x_eval->channels && x_eval->channels <= 65535, "dvc_eb_backward: channel mismatch");
  DVC_REQUIRE(packed_params && scratch && grad_packed, "dvc_eb_backward: null pointer");
  if (grad_x != nullptr && grad_x->ptr != nullptr) CHECK(check_dst(grad_x, h, w, "dvc_eb_backward(grad_x)"));
  hipLaunchKernelGGL(eb_backward_kernel, dim3(EB_CHUNKS, x_eval->channels), dim3(256), 0, (hipStream_t)stream, mk(x_eval),
                     packed_params, mk(grad_lik), mk(grad_x), scratch, batch, h, w);
  CHECK(dvc::check_launch("eb_backward_kernel"));
  hipLaunchKernelGGL(eb_backward_final_kernel, dim3(stream_grid((long long)x_eval->channels * NP, 256)), dim3(256), 0,
                     (hipStream_t)stream, (const float*)scratch, packed_params, grad_packed, x_eval->channels);
  return dvc::check_launch("eb_backward_final_kernel");
}

extern "C" int dvc_dual_prior_pass1_backward(const DvcSrc* grad_yhat1, const DvcDst* grad_y, int32_t batch, int32_t h,
                                             int32_t w, void* stream) {
  DVC_REQUIRE(batch >= 1 && h >= 1 && w >= 1, "dvc_dual_prior_pass1_backward: bad dims");
  CHECK(check_plane(grad_yhat1, h, w, "dvc_dual_prior_pass1_backward"));
  DVC_REQUIRE(grad_yhat1->channels % 2 == 0, "dvc_dual_prior_pass1_backward: odd channel count");
  CHECK(check_dst(grad_y, h, w, "dvc_dual_prior_pass1_backward"));
  const long long total = (long long)batch * grad_yhat1->channels * h * w;
  hipLaunchKernelGGL(dual_prior_pass1_backward_kernel, dim3(stream_grid(total, 256)), dim3(256), 0, (hipStream_t)stream,
                     mk(grad_yhat1), mk(grad_y), batch, h, w);
  return dvc::check_launch("dual_prior_pass1_backward_kernel");
}

extern "C" int dvc_dual_prior_pass2_backward(const DvcSrc* y, const DvcSrc* means, const DvcSrc* scales,
                                             const DvcSrc* spatial, const float* noise, const DvcSrc* grad_yhat,
                                             const DvcSrc* grad_lik, const DvcDst* grad_y, const DvcDst* grad_means,
                                             const DvcDst* grad_scales, const DvcDst* grad_spatial, int32_t batch,
                                             int32_t h, int32_t w, void* stream) {
  DVC_REQUIRE(batch >= 1 && h >= 1 && w >= 1, "dvc_dual_prior_pass2_backward: bad dims");
  CHECK(check_plane(y, h, w, "dvc_dual_prior_pass2_backward(y)"));
  CHECK(check_plane(means, h, w, "dvc_dual_prior_pass2_backward(means)"));
  CHECK(check_plane(scales, h, w, "dvc_dual_prior_pass2_backward(scales)"));
  CHECK(check_plane(spatial, h, w, "dvc_dual_prior_pass2_backward(spatial)"));
  const int C = y->channels;
  DVC_REQUIRE(C % 2 == 0 && means->channels == C && scales->channels == C && spatial->channels == 2 * C,
              "dvc_dual_prior_pass2_backward: channel mismatch");
  if (grad_yhat) { CHECK(check_plane(grad_yhat, h, w, "dvc_dual_prior_pass2_backward(grad_yhat)")); DVC_REQUIRE(grad_yhat->channels == C, "dvc_dual_prior_pass2_backward: grad_yhat channels"); }
  if (grad_lik) { CHECK(check_plane(grad_lik, h, w, "dvc_dual_prior_pass2_backward(grad_lik)")); DVC_REQUIRE(grad_lik->channels == C, "dvc_dual_prior_pass2_backward: grad_lik channels"); }
  CHECK(check_dst(grad_y, h, w, "dvc_dual_prior_pass2_backward(grad_y)"));
  CHECK(check_dst(grad_means, h, w, "dvc_dual_prior_pass2_backward(grad_means)"));
  CHECK(check_dst(grad_scales, h, w, "dvc_dual_prior_pass2_backward(grad_scales)"));
  CHECK(check_dst(grad_spatial, h, w, "dvc_dual_prior_pass2_backward(grad_spatial)"));
  const long long total = (long long)batch * C * h * w;
  hipLaunchKernelGGL(dual_prior_pass2_backward_kernel, dim3(stream_grid(total, 256)), dim3(256), 0, (hipStream_t)stream,
                     mk(y), mk(means), mk(scales), mk(spatial), noise, mk(grad_yhat), mk(grad_lik), mk(grad_y),
                     mk(grad_means), mk(grad_scales), mk(grad_spatial), batch, h, w);
  return dvc::check_launch("dual_prior_pass2_backward_kernel");
}
