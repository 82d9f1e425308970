// Backward (gradient) kernels of the HBM-bound ops of the DMC path, plus the small helpers the
// convolution gradients need around the MFMA kernels: activation-derivative masks,
// PixelShuffle un-shuffle, zero-insertion upsampling for stride-2 data gradients, per-plane dot
// products (bias / gate gradients), and the warp / pooling / bilinear adjoints.
// Same conventions as the forward kernels: one lane per pixel, consecutive lanes = consecutive x.
#include "dvc_common.h"

namespace {

using dvc::stream_grid;

struct Src { const float* p; long long bs, cs; int c; };
struct Dst { float* p; long long bs, cs; };

inline Src mk(const DvcSrc* s) { return Src{s->ptr, s->batch_stride, s->chan_stride, s->channels}; }
inline Dst mk(const DvcDst* d) { return Dst{d->ptr, d->batch_stride, d->chan_stride}; }

int check_plane(const DvcSrc* s, long long plane, const char* who) {
  DVC_REQUIRE(s != nullptr && s->ptr != nullptr && s->channels >= 1, "%s: null/empty source", who);
  DVC_REQUIRE(s->chan_stride >= plane, "%s: channel stride < H*W", who);
  return DVC_OK;
}
#define CHECK_SRC(s, plane, who) do { int rc_ = check_plane((s), (plane), (who)); if (rc_) return rc_; } while (0)
#define CHECK_DST(d, plane, who) \
  DVC_REQUIRE((d) != nullptr && (d)->ptr != nullptr && (d)->chan_stride >= (int64_t)(plane), "%s: bad destination", who)

#define FOREACH(total_expr)                                                             \
  const long long total_ = (total_expr);                                                \
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total_; i += (long long)gridDim.x * 256)

// ------------------------------------------------------------------ conv helpers
// Gradient w.r.t. the conv output before activation, in conv layout (N, C, ho, wo):
//   gpre = g * act'(t),  act'(t) = t > 0 ? 1 : slope   (t = stored activation output),
// g and t given in the stored layout, i.e. PixelShuffle(2)-ed when `ps`.
__global__ __launch_bounds__(256) void act_grad_kernel(Src g, Src t, int has_t, float slope, int ps, float* out,
                                                       int batch, int c, int ho, int wo) {
  FOREACH((long long)batch * c * ho * wo) {
    const int ox = (int)(i % wo);
    long long q = i / wo;
    const int oy = (int)(q % ho); q /= ho;
    const int co = (int)(q % c), n = (int)(q / c);
    long long off_g, off_t;
    if (ps) {
      const long long pix = (long long)(2 * oy + ((co >> 1) & 1)) * (2 * wo) + 2 * ox + (co & 1);
      off_g = n * g.bs + (co >> 2) * g.cs + pix;
      off_t = n * t.bs + (co >> 2) * t.cs + pix;
    } else {
      const long long pix = (long long)oy * wo + ox;
      off_g = n * g.bs + co * g.cs + pix;
      off_t = n * t.bs + co * t.cs + pix;
    }
    float v = g.p[off_g];
    if (has_t) v = t.p[off_t] > 0.f ? v : v * slope;
    out[i] = v;
  }
}

// out = g * (x > 0 ? 1 : slope) (+ plus): the input-activation derivative applied to a data gradient, and
// optionally the gradient that reaches x along another path added in the same pass
__global__ __launch_bounds__(256) void mul_act_grad_kernel(Src g, Src x, Src plus, int has_plus, Dst out, float slope,
                                                           int batch, int h, int w) {
  const long long hw = (long long)h * w;
  FOREACH((long long)batch * g.c * hw) {
    const long long pix = i % hw;
    const long long q = i / hw;
    const int c = (int)(q % g.c), n = (int)(q / g.c);
    const float gv = g.p[n * g.bs + c * g.cs + pix];
    float v = x.p[n * x.bs + c * x.cs + pix] > 0.f ? gv : gv * slope;
    if (has_plus) v += plus.p[n * plus.bs + c * plus.cs + pix];
    out.p[n * out.bs + c * out.cs + pix] = v;
  }
}

// dst (H x W) = src (h x w) placed on the even sites, zeros elsewhere
__global__ __launch_bounds__(256) void zero_upsample2_kernel(Src src, Dst out, int batch, int h, int w, int H, int W) {
  FOREACH((long long)batch * src.c * H * W) {
    const int x = (int)(i % W);
    long long q = i / W;
    const int y = (int)(q % H); q /= H;
    const int c = (int)(q % src.c), n = (int)(q / src.c);
    float v = 0.f;
    if (!(x & 1) && !(y & 1) && (y >> 1) < h && (x >> 1) < w)
      v = src.p[n * src.bs + c * src.cs + (long long)(y >> 1) * w + (x >> 1)];
    out.p[n * out.bs + c * out.cs + (long long)y * W + x] = v;
  }
}

// out = a + b * scale[n, c] (+ c2): the un-fused form of the conv epilogue's residual adds
__global__ __launch_bounds__(256) void add_scaled_kernel(Src a, Src b, const float* scale, Src c2, int has_c2, Dst out,
                                                         int batch, int h, int w) {
  const long long hw = (long long)h * w;
  FOREACH((long long)batch * a.c * hw) {
    const long long pix = i % hw;
    const long long q = i / hw;
    const int c = (int)(q % a.c), n = (int)(q / a.c);
    float bv = b.p[n * b.bs + c * b.cs + pix];
    if (scale != nullptr) bv = __fmul_rn(bv, scale[(long long)n * a.c + c]);
    float v = __fadd_rn(a.p[n * a.bs + c * a.cs + pix], bv);
    if (has_c2) v = __fadd_rn(v, c2.p[n * c2.bs + c * c2.cs + pix]);
    out.p[n * out.bs + c * out.cs + pix] = v;
  }
}

// out = x * scale[n, c]
__global__ __launch_bounds__(256) void scale_planes_kernel(Src x, const float* scale, Dst out, int batch, int h, int w) {
  const long long hw = (long long)h * w;
  FOREACH((long long)batch * x.c * hw) {
    const long long pix = i % hw;
    const long long q = i / hw;
    const int c = (int)(q % x.c), n = (int)(q / x.c);
    out.p[n * out.bs + c * out.cs + pix] = x.p[n * x.bs + c * x.cs + pix] * scale[(long long)n * x.c + c];
  }
}

// ------------------------------------------------------------------ 16-byte variants
// The element-wise kernels above move 4 bytes per lane and pay a 64-bit division per element; when every
// plane is a multiple of 4 floats and all pointers / strides are 16-byte aligned the same work runs on
// float4 units with 32-bit index arithmetic (plane4 = H*W/4 units per plane).  Selected on the host.
__device__ __forceinline__ const float4* plane4_of(const Src& s, unsigned n, unsigned c) {
  return reinterpret_cast<const float4*>(s.p + (long long)n * s.bs + (long long)c * s.cs);
}
__device__ __forceinline__ float4* plane4_of(const Dst& d, unsigned n, unsigned c) {
  return reinterpret_cast<float4*>(d.p + (long long)n * d.bs + (long long)c * d.cs);
}
#define FOREACH4(total_expr)                                                  \
  const unsigned total_ = (total_expr);                                       \
  for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < total_; i += gridDim.x * 256u)

__global__ __launch_bounds__(256) void act_grad4_kernel(Src g, Src t, int has_t, float slope, float4* out, unsigned chans,
                                                        unsigned plane4, unsigned total) {
  FOREACH4(total) {
    const unsigned p4 = i % plane4, q = i / plane4, c = q % chans, n = q / chans;
    float4 v = plane4_of(g, n, c)[p4];
    if (has_t) {
      const float4 a = plane4_of(t, n, c)[p4];
      v.x = a.x > 0.f ? v.x : v.x * slope; v.y = a.y > 0.f ? v.y : v.y * slope;
      v.z = a.z > 0.f ? v.z : v.z * slope; v.w = a.w > 0.f ? v.w : v.w * slope;
    }
    out[i] = v;
  }
}

__global__ __launch_bounds__(256) void mul_act_grad4_kernel(Src g, Src x, Src plus, int has_plus, Dst out, float slope,
                                                            unsigned chans, unsigned plane4, unsigned total) {
  FOREACH4(total) {
    const unsigned p4 = i % plane4, q = i / plane4, c = q % chans, n = q / chans;
    float4 v = plane4_of(g, n, c)[p4];
    const float4 a = plane4_of(x, n, c)[p4];
    v.x = a.x > 0.f ? v.x : v.x * slope; v.y = a.y > 0.f ? v.y : v.y * slope;
    v.z = a.z > 0.f ? v.z : v.z * slope; v.w = a.w > 0.f ? v.w : v.w * slope;
    if (has_plus) {
      const float4 e = plane4_of(plus, n, c)[p4];
      v.x += e.x; v.y += e.y; v.z += e.z; v.w += e.w;
    }
    plane4_of(out, n, c)[p4] = v;
  }
}

__global__ __launch_bounds__(256) void add_scaled4_kernel(Src a, Src b, const float* scale, Src c2, int has_c2, Dst out,
                                                          unsigned chans, unsigned plane4, unsigned total) {
  FOREACH4(total) {
    const unsigned p4 = i % plane4, q = i / plane4, c = q % chans, n = q / chans;
    float4 bv = plane4_of(b, n, c)[p4];
    if (scale != nullptr) {
      const float sc = scale[q];
      bv.x = __fmul_rn(bv.x, sc); bv.y = __fmul_rn(bv.y, sc); bv.z = __fmul_rn(bv.z, sc); bv.w = __fmul_rn(bv.w, sc);
    }
    const float4 av = plane4_of(a, n, c)[p4];
    float4 v = make_float4(__fadd_rn(av.x, bv.x), __fadd_rn(av.y, bv.y), __fadd_rn(av.z, bv.z), __fadd_rn(av.w, bv.w));
    if (has_c2) {
      const float4 cv = plane4_of(c2, n, c)[p4];
      v.x = __fadd_rn(v.x, cv.x); v.y = __fadd_rn(v.y, cv.y); v.z = __fadd_rn(v.z, cv.z); v.w = __fadd_rn(v.w, cv.w);
    }
    plane4_of(out, n, c)[p4] = v;
  }
}

__global__ __launch_bounds__(256) void scale_planes4_kernel(Src x, const float* scale, Dst out, unsigned chans,
                                                            unsigned plane4, unsigned total) {
  FOREACH4(total) {
    const unsigned p4 = i % plane4, q = i / plane4, c = q % chans, n = q / chans;
    float4 v = plane4_of(x, n, c)[p4];
    const float sc = scale[q];
    v.x *= sc; v.y *= sc; v.z *= sc; v.w *= sc;
    plane4_of(out, n, c)[p4] = v;
  }
}

// zero insertion, one float4 of the output row per lane: (s0, 0, s1, 0) on even rows, zeros on odd rows
// (needs W % 4 == 0 and 2 w == W, 2 h == H; the source is read as float2)
__global__ __launch_bounds__(256) void zero_upsample2_4_kernel(Src src, Dst out, unsigned chans, unsigned w, unsigned H,
                                                               unsigned W4, unsigned total) {
  FOREACH4(total) {
    const unsigned x4 = i % W4;
    unsigned q = i / W4;
    const unsigned y = q % H;
    q /= H;
    const unsigned c = q % chans, n = q / chans;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (!(y & 1)) {
      const float2 s2 = *reinterpret_cast<const float2*>(src.p + (long long)n * src.bs + (long long)c * src.cs +
                                                         (long long)(y >> 1) * w + 2 * x4);
      v.x = s2.x;
      v.z = s2.y;
    }
    plane4_of(out, n, c)[(unsigned long long)y * W4 + x4] = v;
  }
}

// host side: can this operand be addressed in float4 units?
inline bool al16(const void* p, long long bs, long long cs) {
  return reinterpret_cast<uintptr_t>(p) % 16 == 0 && bs % 4 == 0 && cs % 4 == 0;
}
inline bool al16(const DvcSrc* s) { return s == nullptr || al16(s->ptr, s->batch_stride, s->chan_stride); }
inline bool al16(const DvcDst* d) { return al16(d->ptr, d->batch_stride, d->chan_stride); }

// ------------------------------------------------------------------ plane dot products
constexpr int SLICES = 64;

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

__global__ __launch_bounds__(256) void plane_dot_partial_kernel(Src a, Src b, int has_b, long long hw, float* partial) {
  __shared__ float red[4];
  const int slice = blockIdx.x, c = blockIdx.y, n = blockIdx.z;
  const int nslices = gridDim.x;   // <= SLICES, chosen by the host from the plane size
  const long long per = (hw + nslices - 1) / nslices;
  const long long s0 = slice * per, s1 = s0 + per < hw ? s0 + per : hw;
  const float* pa = a.p + n * a.bs + c * a.cs;
  const float* pb = has_b ? b.p + n * b.bs + c * b.cs : nullptr;
  float s = 0.f;
  if (has_b) for (long long i = s0 + threadIdx.x; i < s1; i += 256) s += pa[i] * pb[i];
  else for (long long i = s0 + threadIdx.x; i < s1; i += 256) s += pa[i];
  s = block_sum_256(s, red);
  if (threadIdx.x == 0) partial[((long long)n * a.c + c) * nslices + slice] = s;
}

// out[n*C + c] (sum_batch = 0) or out[c] = sum over n (sum_batch = 1), fixed order
__global__ __launch_bounds__(256) void plane_dot_final_kernel(const float* partial, int batch, int channels,
                                                              int sum_batch, int nslices, float* out) {
  const int total = sum_batch ? channels : batch * channels;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
    float s = 0.f;
    if (sum_batch) {
      for (int n = 0; n < batch; ++n) {
        const float* pp = partial + ((long long)n * channels + i) * nslices;
        float sn = 0.f;
        for (int k = 0; k < nslices; ++k) sn += pp[k];
        s += sn;
      }
    } else {
      const float* pp = partial + (long long)i * nslices;
      for (int k = 0; k < nslices; ++k) s += pp[k];
    }
    out[i] = s;
  }
}

// ------------------------------------------------------------------ warp adjoint
__device__ __forceinline__ float linspace_pm1(int idx, int steps) {
  const float step = 2.0f / (float)(steps - 1);
  const int halfway = steps / 2;
  return idx < halfway ? -1.0f + step * (float)idx : 1.0f - step * (float)(steps - idx - 1);
}

// Adjoint of the backward warp (same coordinate arithmetic as the forward kernel).  gsrc must be
// zeroed by the caller; it is accumulated with float atomics (several output pixels may sample
// the same source pixel).  gflow follows grid_sample's rule: no gradient through a coordinate
// that was clipped to the border (position <= 0 or >= size - 1).
//
// A workgroup takes an 8 x 32 tile of output pixels.  The sampling positions and the four bilinear
// weights of a pixel are the same for every channel, and the flow is smooth: the 2 x 2 neighbourhoods
// of a tile fall into a small window of the source plane.  When that window fits the LDS tile (it
// does unless the flow varies by more than ~50 pixels inside the tile), the channels go through LDS:
// the source window is loaded with coalesced reads, the gradient is scattered with LDS atomics, and the
// window is flushed with one global atomic per touched element -- instead of four scattered global
// atomics and four scattered loads per pixel and channel.  The LDS holds sums and source values of four
// channels of a window of up to 1280 elements (smooth motion), of two of up to 2560, of one of up to 5120, or
// the sums alone of one channel of up to 10240 with the source values gathered from memory (a noisy motion
// field, as the decoded motion of an untrained codec is: 84 % of the tiles of a 256 x 256 crop exceeded 1280
// elements there, 43 % 5120, 3 % 10240, and the launch took 18.7 ms on scattered global atomics).  Tiles
// whose window still does not fit take those directly.
//
// The LDS sums are 32-bit fixed point: a float add in LDS is carried out lane by lane (a wave's ds_add_f32
// took ~160 cycles here, four per pixel and channel: 60 % of the kernel), an integer add is not.  Per tile
// and channel the scale is the power of two that brings the largest |gradient| of the tile's 256 pixels to
// [2^21, 2^22): an element collects at most one tap of each pixel, so a sum stays below 2^30, and a
// contribution is rounded at 2^-22 of that maximum.  Integer sums do not depend on the order of arrival.  A
// tile whose gradients are not all finite keeps float adds for that channel, so that NaN / Inf still propagate.
struct WarpTap {
  int o_nw;            // offset of the north-west tap in the source plane
  bool in_e, in_s;     // east / south taps exist
  float ee, we, ss, ns;
  float mx, my, sx, sy;
  int ix, iy;
};

__device__ __forceinline__ WarpTap warp_tap(const Src& flow, int n, int pix, int x, int y, int h, int w) {
  WarpTap t;
  const float* fp = flow.p + n * flow.bs + pix;
  const float u = fp[0], v = fp[flow.cs];
  t.sx = (float)((w - 1.0) / 2.0);
  t.sy = (float)((h - 1.0) / 2.0);
  float gx = linspace_pm1(x, w) + u / t.sx;
  float gy = linspace_pm1(y, h) + v / t.sy;
  if (w == 1) gx = -1.0f + u / t.sx;
  if (h == 1) gy = -1.0f + v / t.sy;
  float fx = (gx + 1.0f) * t.sx, fy = (gy + 1.0f) * t.sy;
  t.mx = (fx <= 0.f || fx >= (float)(w - 1)) ? 0.f : 1.f;
  t.my = (fy <= 0.f || fy >= (float)(h - 1)) ? 0.f : 1.f;
  fx = fminf((float)(w - 1), fmaxf(fx, 0.0f));
  fy = fminf((float)(h - 1), fmaxf(fy, 0.0f));
  const float xw = floorf(fx), yn = floorf(fy);
  t.we = fx - xw;
  t.ee = 1.0f - t.we;
  t.ns = fy - yn;
  t.ss = 1.0f - t.ns;
  t.ix = (int)xw;
  t.iy = (int)yn;
  t.in_e = t.ix + 1 < w;
  t.in_s = t.iy + 1 < h;
  t.o_nw = t.iy * w + t.ix;
  return t;
}

constexpr int WB_TH = 8, WB_TW = 32, WB_CH = 4, WB_WIN = 1280;   // window elements per channel (e.g. 24 x 48)

__global__ __launch_bounds__(256) void flow_warp_backward_kernel(Src src, Src flow, Src gout, Dst gsrc, int want_gsrc,
                                                                 Dst gflow, int want_gflow, int h, int w) {
  constexpr int WB_FLOATS = WB_CH * WB_WIN;
  // [channel of the pass][window element] fixed-point sums (float bits for a channel whose gradients are not
  // finite), then the source window of the same channels -- or, for a window of more than WB_FLOATS elements,
  // one channel's sums alone (the source values are then gathered from memory)
  __shared__ int s_buf[2 * WB_FLOATS];
  __shared__ float s_max[WB_CH][4];    // per channel of the pass and wave: largest |gradient|
  __shared__ int s_box[4];   // min x, min y, max x, max y of the taps of the tile
  const int tid = threadIdx.x;
  const int n = blockIdx.z;
  const int x = blockIdx.x * WB_TW + (tid & 31), y = blockIdx.y * WB_TH + (tid >> 5);
  const bool valid = x < w && y < h;
  const int pix = valid ? y * w + x : 0;
  if (tid == 0) {
    s_box[0] = s_box[1] = 0x7fffffff;
    s_box[2] = s_box[3] = -1;
  }
  __syncthreads();
  WarpTap t = warp_tap(flow, n, pix, valid ? x : 0, valid ? y : 0, h, w);
  if (valid) {
    atomicMin(&s_box[0], t.ix);
    atomicMin(&s_box[1], t.iy);
    atomicMax(&s_box[2], t.ix + (t.in_e ? 1 : 0));
    atomicMax(&s_box[3], t.iy + (t.in_s ? 1 : 0));
  }
  __syncthreads();
  const int wx0 = s_box[0], wy0 = s_box[1];
  const int ww = s_box[2] - wx0 + 1, wh = s_box[3] - wy0 + 1;
  const int wsize = ww * wh;
  float gix = 0.f, giy = 0.f;

  if (wsize <= 2 * WB_FLOATS) {   // workgroup-uniform
    const int l_nw = (t.iy - wy0) * ww + (t.ix - wx0);
    const bool stage_src = want_gflow && wsize <= WB_FLOATS;
    const int cpp = wsize <= WB_FLOATS ? min(WB_CH, WB_FLOATS / wsize) : 1;   // channels per pass
    int* const s_acc = s_buf;
    float* const s_src = reinterpret_cast<float*>(s_buf + cpp * wsize);
    // window elements tid, tid + 256, ... of a channel as (row, column), stepped without divisions
    const int step_y = 256 / ww, step_x = 256 - step_y * ww;
    const int ey0 = tid / ww, ex0 = tid - ey0 * ww;
    const long long win_org = (long long)wy0 * w + wx0;
    for (int c0 = 0; c0 < gout.c; c0 += cpp) {
      const int nch = min(cpp, gout.c - c0);
      float g[WB_CH];
#pragma unroll
      for (int k = 0; k < WB_CH; ++k) {
        g[k] = (valid && k < nch) ? gout.p[n * gout.bs + (c0 + k) * gout.cs + pix] : 0.f;
        float m = fabsf(g[k]);
        if (!(m <= 3.0e38f)) m = __builtin_inff();   // NaN / Inf: marks the channel
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
        if ((tid & 63) == 0) s_max[k][tid >> 6] = m;
      }
      for (int k = 0; k < nch; ++k) {
        const float* const sp = src.p + n * src.bs + (c0 + k) * src.cs + win_org;
        int ey = ey0, ex = ex0;
        for (int e = tid; e < wsize; e += 256) {
          s_acc[k * wsize + e] = 0;
          if (stage_src) s_src[k * wsize + e] = sp[ey * w + ex];
          ex += step_x; ey += step_y;
          if (ex >= ww) { ex -= ww; ++ey; }
        }
      }
      __syncthreads();
      float unscale[WB_CH];   // 0: float sums
#pragma unroll
      for (int k = 0; k < WB_CH; ++k) {
        const float m = fmaxf(fmaxf(s_max[k][0], s_max[k][1]), fmaxf(s_max[k][2], s_max[k][3]));
        int ex2 = 0;
        (void)frexpf(m, &ex2);                      // m = f 2^ex2, f in [0.5, 1)
        ex2 = max(-100, ex2);                       // (a tile of vanishing gradients: keeps the scale finite)
        unscale[k] = m <= 3.0e38f ? ldexpf(1.0f, ex2 - 22) : 0.f;
        if (valid && k < nch) {
          if (want_gsrc) {
            int* a = s_acc + k * wsize + l_nw;
            const float c_nw = g[k] * (t.ss * t.ee), c_ne = g[k] * (t.ss * t.we);
            const float c_sw = g[k] * (t.ns * t.ee), c_se = g[k] * (t.ns * t.we);
            if (unscale[k] != 0.f) {
              const float sc = ldexpf(1.0f, 22 - ex2);
              atomicAdd(a, __float2int_rn(c_nw * sc));
              if (t.in_e) atomicAdd(a + 1, __float2int_rn(c_ne * sc));
              if (t.in_s) atomicAdd(a + ww, __float2int_rn(c_sw * sc));
              if (t.in_e && t.in_s) atomicAdd(a + ww + 1, __float2int_rn(c_se * sc));
            } else {
              float* af = reinterpret_cast<float*>(a);
              atomicAdd(af, c_nw);
              if (t.in_e) atomicAdd(af + 1, c_ne);
              if (t.in_s) atomicAdd(af + ww, c_sw);
              if (t.in_e && t.in_s) atomicAdd(af + ww + 1, c_se);
            }
          }
          if (want_gflow) {
            float nw, ne, sw, se;
            if (stage_src) {   // workgroup-uniform
              const float* sp = s_src + k * wsize + l_nw;
              nw = sp[0];
              ne = t.in_e ? sp[1] : 0.f;
              sw = t.in_s ? sp[ww] : 0.f;
              se = (t.in_e && t.in_s) ? sp[ww + 1] : 0.f;
            } else {
              const float* sp = src.p + n * src.bs + (c0 + k) * src.cs + t.o_nw;
              nw = sp[0];
              ne = t.in_e ? sp[1] : 0.f;
              sw = t.in_s ? sp[w] : 0.f;
              se = (t.in_e && t.in_s) ? sp[w + 1] : 0.f;
            }
            gix += g[k] * ((ne - nw) * t.ss + (se - sw) * t.ns);
            giy += g[k] * ((sw - nw) * t.ee + (se - ne) * t.we);
          }
        }
      }
      __syncthreads();
      if (want_gsrc) {
#pragma unroll
        for (int k = 0; k < WB_CH; ++k) {
          if (k >= nch) break;
          float* const gp = gsrc.p + n * gsrc.bs + (c0 + k) * gsrc.cs + win_org;
          const float us = unscale[k];
          int ey = ey0, ex = ex0;
          for (int e = tid; e < wsize; e += 256) {
            const int q = s_acc[k * wsize + e];
            if (q != 0) atomicAdd(gp + ey * w + ex, us != 0.f ? (float)q * us : __int_as_float(q));
            ex += step_x; ey += step_y;
            if (ex >= ww) { ex -= ww; ++ey; }
          }
        }
      }
      __syncthreads();
    }
  } else if (valid) {
    const int o_nw = t.o_nw, o_ne = o_nw + 1, o_sw = o_nw + w, o_se = o_nw + w + 1;
    for (int c = 0; c < gout.c; ++c) {
      const float g = gout.p[n * gout.bs + c * gout.cs + pix];
      if (want_gsrc) {
        float* gs = gsrc.p + n * gsrc.bs + c * gsrc.cs;
        atomicAdd(gs + o_nw, g * (t.ss * t.ee));
        if (t.in_e) atomicAdd(gs + o_ne, g * (t.ss * t.we));
        if (t.in_s) atomicAdd(gs + o_sw, g * (t.ns * t.ee));
        if (t.in_e && t.in_s) atomicAdd(gs + o_se, g * (t.ns * t.we));
      }
      if (want_gflow) {
        const float* sp = src.p + n * src.bs + c * src.cs;
        const float nw = sp[o_nw];
        const float ne = t.in_e ? sp[o_ne] : 0.f;
        const float sw = t.in_s ? sp[o_sw] : 0.f;
        const float se = (t.in_e && t.in_s) ? sp[o_se] : 0.f;
        gix += g * ((ne - nw) * t.ss + (se - sw) * t.ns);
        giy += g * ((sw - nw) * t.ee + (se - ne) * t.we);
      }
    }
  }
  if (want_gflow && valid) {
    float* gf = gflow.p + n * gflow.bs + pix;
    gf[0] = (gix * t.sx * t.mx) / t.sx;
    gf[gflow.cs] = (giy * t.sy * t.my) / t.sy;
  }
}

// ------------------------------------------------------------------ pooling / bilinear adjoints
__global__ __launch_bounds__(256) void avg_pool2_backward_kernel(Src g, Dst out, int batch, int h, int w, float scale) {
  const int ho = h / 2, wo = w / 2;
  FOREACH((long long)batch * g.c * h * w) {
    const int x = (int)(i % w);
    long long q = i / w;
    const int y = (int)(q % h); q /= h;
    const int c = (int)(q % g.c), n = (int)(q / g.c);
    float v = 0.f;
    if ((y >> 1) < ho && (x >> 1) < wo) v = 0.25f * scale * g.p[n * g.bs + c * g.cs + (long long)(y >> 1) * wo + (x >> 1)];
    out.p[n * out.bs + c * out.cs + (long long)y * w + x] = v;
  }
}

// gradient goes to the first maximum in scan order (0,0), (0,1), (1,0), (1,1)
__global__ __launch_bounds__(256) void max_pool2_backward_kernel(Src xin, Src g, Dst out, int batch, int h, int w) {
  const int ho = h / 2, wo = w / 2;
  FOREACH((long long)batch * g.c * h * w) {
    const int x = (int)(i % w);
    long long q = i / w;
    const int y = (int)(q % h); q /= h;
    const int c = (int)(q % g.c), n = (int)(q / g.c);
    float v = 0.f;
    const int oy = y >> 1, ox = x >> 1;
    if (oy < ho && ox < wo) {
      const float* p = xin.p + n * xin.bs + c * xin.cs + (long long)(2 * oy) * w + 2 * ox;
      int best = 0;
      float bv = p[0];
      if (p[1] > bv) { bv = p[1]; best = 1; }
      if (p[w] > bv) { bv = p[w]; best = 2; }
      if (p[w + 1] > bv) { bv = p[w + 1]; best = 3; }
      if (best == ((y & 1) << 1 | (x & 1))) v = g.p[n * g.bs + c * g.cs + (long long)oy * wo + ox];
    }
    out.p[n * out.bs + c * out.cs + (long long)y * w + x] = v;
  }
}

__device__ __forceinline__ void up2_coord(int o, int in, int& i0, int& i1, float& l0, float& l1) {
  float s = 0.5f * ((float)o + 0.5f) - 0.5f;
  s = s < 0.f ? 0.f : s;
  i0 = (int)s;
  i1 = i0 + (i0 < in - 1 ? 1 : 0);
  l1 = s - (float)i0;
  l0 = 1.0f - l1;
}

// weight with which output coordinate o of the x2 upsampling reads input coordinate i
__device__ __forceinline__ float up2_weight(int o, int i, int in) {
  int i0, i1; float l0, l1;
  up2_coord(o, in, i0, i1, l0, l1);
  return (i0 == i ? l0 : 0.f) + (i1 == i ? l1 : 0.f);
}

// gather form of the adjoint: input pixel (y, x) is read by output rows 2y-1 .. 2y+2 (cols likewise)
__global__ __launch_bounds__(256) void bilinear_up2_backward_kernel(Src g, Dst out, int batch, int h, int w, float scale) {
  const int ho = 2 * h, wo = 2 * w;
  FOREACH((long long)batch * g.c * h * w) {
    const int x = (int)(i % w);
    long long q = i / w;
    const int y = (int)(q % h); q /= h;
    const int c = (int)(q % g.c), n = (int)(q / g.c);
    const float* gp = g.p + n * g.bs + c * g.cs;
    float s = 0.f;
    for (int oy = 2 * y - 1; oy <= 2 * y + 2; ++oy) {
      if (oy < 0 || oy >= ho) continue;
      const float wy = up2_weight(oy, y, h);
      if (wy == 0.f) continue;
      float r = 0.f;
      for (int ox = 2 * x - 1; ox <= 2 * x + 2; ++ox) {
        if (ox < 0 || ox >= wo) continue;
        r += up2_weight(ox, x, w) * gp[(long long)oy * wo + ox];
      }
      s += wy * r;
    }
    out.p[n * out.bs + c * out.cs + (long long)y * w + x] = s * scale;
  }
}

// ------------------------------------------------------------------ loss reductions' adjoints
// bits[n] = sum -log2(lik[n])  =>  dlik = -g[n] / (ln 2 * lik)
__global__ __launch_bounds__(256) void likelihood_bits_backward_kernel(Src lik, const float* gbits, Dst out, int batch,
                                                                       int h, int w) {
  const long long hw = (long long)h * w;
  FOREACH((long long)batch * lik.c * hw) {
    const long long pix = i % hw;
    const long long q = i / hw;
    const int c = (int)(q % lik.c), n = (int)(q / lik.c);
    out.p[n * out.bs + c * out.cs + pix] = -gbits[n] / (0.69314718055994530942f * lik.p[n * lik.bs + c * lik.cs + pix]);
  }
}

// sse[n] = sum (a - b)^2  =>  da = 2 g[n] (a - b)   (db = -da)
__global__ __launch_bounds__(256) void sq_err_backward_kernel(Src a, Src b, const float* gsse, Dst out, int batch, int h,
                                                              int w) {
  const long long hw = (long long)h * w;
  FOREACH((long long)batch * a.c * hw) {
    const long long pix = i % hw;
    const long long q = i / hw;
    const int c = (int)(q % a.c), n = (int)(q / a.c);
    out.p[n * out.bs + c * out.cs + pix] =
        2.0f * gsse[n] * (a.p[n * a.bs + c * a.cs + pix] - b.p[n * b.bs + c * b.cs + pix]);
  }
}

}  // namespace

#define LAUNCH_1D(kernel, total, stream, ...) \
  hipLaunchKernelGGL(kernel, dim3(stream_grid((total), 256)), dim3(256), 0, (hipStream_t)(stream), __VA_ARGS__)

extern "C" int dvc_act_grad(const DvcSrc* grad, const DvcSrc* act_out, float slope, int32_t pixel_shuffle,
                            float* grad_pre, int32_t batch, int32_t channels, int32_t ho, int32_t wo, void* stream) {
  DVC_REQUIRE(batch >= 1 && channels >= 1 && ho >= 1 && wo >= 1 && grad_pre != nullptr, "dvc_act_grad: bad arguments");
  DVC_REQUIRE(!pixel_shuffle || channels % 4 == 0, "dvc_act_grad: pixel_shuffle needs C %% 4 == 0");
  const long long plane = (long long)ho * wo * (pixel_shuffle ? 4 : 1);
  const int stored_c = pixel_shuffle ? channels / 4 : channels;
  CHECK_SRC(grad, plane, "dvc_act_grad(grad)");
  DVC_REQUIRE(grad->channels == stored_c, "dvc_act_grad: grad has %d channels, expected %d", grad->channels, stored_c);
  if (act_out != nullptr) {
    CHECK_SRC(act_out, plane, "dvc_act_grad(act_out)");
    DVC_REQUIRE(act_out->channels == stored_c, "dvc_act_grad: act_out channel mismatch");
  }
  const long long total = (long long)batch * channels * ho * wo;
  if (!pixel_shuffle && plane % 4 == 0 && total / 4 < (1LL << 31) && al16(grad) && al16(act_out) &&
      reinterpret_cast<uintptr_t>(grad_pre) % 16 == 0) {
    LAUNCH_1D(act_grad4_kernel, total / 4, stream, mk(grad), act_out ? mk(act_out) : Src{nullptr, 0, 0, 0}, act_out ? 1 : 0,
              slope, reinterpret_cast<float4*>(grad_pre), (unsigned)channels, (unsigned)(plane / 4), (unsigned)(total / 4));
    return dvc::check_launch("act_grad4_kernel");
  }
  LAUNCH_1D(act_grad_kernel, total, stream, mk(grad), act_out ? mk(act_out) : Src{nullptr, 0, 0, 0}, act_out ? 1 : 0,
            slope, pixel_shuffle ? 1 : 0, grad_pre, batch, channels, ho, wo);
  return dvc::check_launch("act_grad_kernel");
}

extern "C" int dvc_mul_act_grad_add(const DvcSrc* grad, const DvcSrc* x, float slope, const DvcSrc* plus,
                                    const DvcDst* out, int32_t batch, int32_t h, int32_t w, void* stream) {
  DVC_REQUIRE(batch >= 1 && h >= 1 && w >= 1, "dvc_mul_act_grad: bad dims");
  const long long plane = (long long)h * w;
  CHECK_SRC(grad, plane, "dvc_mul_act_grad(grad)");
  CHECK_SRC(x, plane, "dvc_mul_act_grad(x)");
  DVC_REQUIRE(x->channels == grad->channels, "dvc_mul_act_grad: channel mismatch");
  if (plus != nullptr) {
    CHECK_SRC(plus, plane, "dvc_mul_act_grad(plus)");
    DVC_REQUIRE(plus->channels == grad->channels, "dvc_mul_act_grad: channel mismatch (plus)");
  }
  CHECK_DST(out, plane, "dvc_mul_act_grad");
  const Src pl = plus != nullptr ? mk(plus) : mk(grad);
  const int has_plus = plus != nullptr ? 1 : 0;
  {
    const long long total = (long long)batch * grad->channels * plane;
    if (plane % 4 == 0 && total / 4 < (1LL << 31) && al16(grad) && al16(x) && al16(out) && (plus == nullptr || al16(plus))) {
      LAUNCH_1D(mul_act_grad4_kernel, total / 4, stream, mk(grad), mk(x), pl, has_plus, mk(out), slope,
                (unsigned)grad->channels, (unsigned)(plane / 4), (unsigned)(total / 4));
      return dvc::check_launch("mul_act_grad4_kernel");
    }
  }
  LAUNCH_1D(mul_act_grad_kernel, (long long)batch * grad->channels * plane, stream, mk(grad), mk(x), pl, has_plus, mk(out),
            slope, batch, h, w);
  return dvc::check_launch("mul_act_grad_kernel");
}

extern "C" int dvc_mul_act_grad(const DvcSrc* grad, const DvcSrc* x, float slope, const DvcDst* out, int32_t batch,
                                int32_t h, int32_t w, void* stream) {
  return dvc_mul_act_grad_add(grad, x, slope, nullptr, out, batch, h, w, stream);
}

extern "C" int dvc_zero_upsample2(const DvcSrc* src, const DvcDst* out, int32_t batch, int32_t h, int32_t w,
                                  int32_t out_h, int32_t out_w, void* stream) {
  DVC_REQUIRE(batch >= 1 && h >= 1 && w >= 1 && out_h >= 2 * h - 1 && out_w >= 2 * w - 1, "dvc_zero_upsample2: bad dims");
  CHECK_SRC(src, (long long)h * w, "dvc_zero_upsample2");
  CHECK_DST(out, (long long)out_h * out_w, "dvc_zero_upsample2");
  {
    const long long total = (long long)batch * src->channels * out_h * out_w;
    if (out_w == 2 * w && out_h == 2 * h && out_w % 4 == 0 && total / 4 < (1LL << 31) && al16(out) &&
        reinterpret_cast<uintptr_t>(src->ptr) % 8 == 0 && src->batch_stride % 2 == 0 && src->chan_stride % 2 == 0) {
      LAUNCH_1D(zero_upsample2_4_kernel, total / 4, stream, mk(src), mk(out), (unsigned)src->channels, (unsigned)w,
                (unsigned)out_h, (unsigned)(out_w / 4), (unsigned)(total / 4));
      return dvc::check_launch("zero_upsample2_4_kernel");
    }
  }
  LAUNCH_1D(zero_upsample2_kernel, (long long)batch * src->channels * out_h * out_w, stream, mk(src), mk(out), batch,
            h, w, out_h, out_w);
  return dvc::check_launch("zero_upsample2_kernel");
}

extern "C" int dvc_add_scaled(const DvcSrc* a, const DvcSrc* b, const float* scale, const DvcSrc* c, const DvcDst* out,
                              int32_t batch, int32_t h, int32_t w, void* stream) {
  DVC_REQUIRE(batch >= 1 && h >= 1 && w >= 1, "dvc_add_scaled: bad dims");
  const long long plane = (long long)h * w;
  CHECK_SRC(a, plane, "dvc_add_scaled(a)");
  CHECK_SRC(b, plane, "dvc_add_scaled(b)");
  DVC_REQUIRE(b->channels == a->channels, "dvc_add_scaled: channel mismatch");
  if (c != nullptr) {
    CHECK_SRC(c, plane, "dvc_add_scaled(c)");
    DVC_REQUIRE(c->channels == a->channels, "dvc_add_scaled: channel mismatch");
  }
  CHECK_DST(out, plane, "dvc_add_scaled");
  {
    const long long total = (long long)batch * a->channels * plane;
    if (plane % 4 == 0 && total / 4 < (1LL << 31) && al16(a) && al16(b) && al16(c) && al16(out)) {
      LAUNCH_1D(add_scaled4_kernel, total / 4, stream, mk(a), mk(b), scale, c ? mk(c) : Src{nullptr, 0, 0, 0}, c ? 1 : 0,
                mk(out), (unsigned)a->channels, (unsigned)(plane / 4), (unsigned)(total / 4));
      return dvc::check_launch("add_scaled4_kernel");
    }
  }
  LAUNCH_1D(add_scaled_kernel, (long long)batch * a->channels * plane, stream, mk(a), mk(b), scale,
            c ? mk(c) : Src{nullptr, 0, 0, 0}, c ? 1 : 0, mk(out), batch, h, w);
  return dvc::check_launch("add_scaled_kernel");
}

extern "C" int dvc_scale_planes(const DvcSrc* x, const float* scale, const DvcDst* out, int32_t batch, int32_t h,
                                int32_t w, void* stream) {
  DVC_REQUIRE(batch >= 1 && h >= 1 && w >= 1 && scale != nullptr, "dvc_scale_planes: bad arguments");
  const long long plane = (long long)h * w;
  CHECK_SRC(x, plane, "dvc_scale_planes");
  CHECK_DST(out, plane, "dvc_scale_planes");
  {
    const long long total = (long long)batch * x->channels * plane;
    if (plane % 4 == 0 && total / 4 < (1LL << 31) && al16(x) && al16(out)) {
      LAUNCH_1D(scale_planes4_kernel, total / 4, stream, mk(x), scale, mk(out), (unsigned)x->channels, (unsigned)(plane / 4),
                (unsigned)(total / 4));
      return dvc::check_launch("scale_planes4_kernel");
    }
  }
  LAUNCH_1D(scale_planes_kernel, (long long)batch * x->channels * plane, stream, mk(x), scale, mk(out), batch, h, w);
  return dvc::check_launch("scale_planes_kernel");
}

extern "C" int64_t dvc_plane_dot_scratch_floats(int32_t batch, int32_t channels) {
  return (int64_t)batch * channels * SLICES;
}

extern "C" int dvc_plane_dot(const DvcSrc* a, const DvcSrc* b, int32_t batch, int32_t h, int32_t w, int32_t sum_batch,
                             float* scratch, float* out, void* stream) {
  DVC_REQUIRE(batch >= 1 && batch <= 65535 && h >= 1 && w >= 1 && scratch && out, "dvc_plane_dot: bad arguments");
  const long long plane = (long long)h * w;
  CHECK_SRC(a, plane, "dvc_plane_dot(a)");
  DVC_REQUIRE(a->channels <= 65535, "dvc_plane_dot: too many channels");
  if (b != nullptr) {
    CHECK_SRC(b, plane, "dvc_plane_dot(b)");
    DVC_REQUIRE(b->channels == a->channels, "dvc_plane_dot: channel mismatch");
  }
  // about 4096 elements per block, so that small planes are not shredded into tiny blocks
  int nslices = (int)((plane + 4095) / 4096);
  nslices = nslices < 1 ? 1 : (nslices > SLICES ? SLICES : nslices);
  hipLaunchKernelGGL(plane_dot_partial_kernel, dim3(nslices, a->channels, batch), dim3(256), 0, (hipStream_t)stream,
                     mk(a), b ? mk(b) : Src{nullptr, 0, 0, 0}, b ? 1 : 0, plane, scratch);
  int rc = dvc::check_launch("plane_dot_partial_kernel");
  if (rc) return rc;
  const int total = sum_batch ? a->channels : batch * a->channels;
  LAUNCH_1D(plane_dot_final_kernel, total, stream, (const float*)scratch, batch, a->channels, sum_batch ? 1 : 0, nslices,
            out);
  return dvc::check_launch("plane_dot_final_kernel");
}

extern "C" int dvc_flow_warp_backward(const DvcSrc* src, const DvcSrc* flow, const DvcSrc* grad_out,
                                      const DvcDst* grad_src, const DvcDst* grad_flow, int32_t batch, int32_t h,
                                      int32_t w, void* stream) {
  DVC_REQUIRE(batch >= 1 && h >= 1 && w >= 1, "dvc_flow_warp_backward: bad dims");
  const long long plane = (long long)h * w;
  DVC_REQUIRE(plane < (1LL << 30), "dvc_flow_warp_backward: plane too large");
  CHECK_SRC(src, plane, "dvc_flow_warp_backward(src)");
  CHECK_SRC(flow, plane, "dvc_flow_warp_backward(flow)");
  CHECK_SRC(grad_out, plane, "dvc_flow_warp_backward(grad_out)");
  DVC_REQUIRE(flow->channels == 2 && grad_out->channels == src->channels, "dvc_flow_warp_backward: channel mismatch");
  const bool ws = grad_src != nullptr && grad_src->ptr != nullptr, wf = grad_flow != nullptr && grad_flow->ptr != nullptr;
  DVC_REQUIRE(ws || wf, "dvc_flow_warp_backward: nothing to compute");
  if (ws) CHECK_DST(grad_src, plane, "dvc_flow_warp_backward(grad_src)");
  if (wf) CHECK_DST(grad_flow, plane, "dvc_flow_warp_backward(grad_flow)");
  DVC_REQUIRE(batch <= 65535, "dvc_flow_warp_backward: batch too large");
  hipLaunchKernelGGL(flow_warp_backward_kernel, dim3((w + WB_TW - 1) / WB_TW, (h + WB_TH - 1) / WB_TH, batch), dim3(256), 0,
                     (hipStream_t)stream, mk(src), mk(flow), mk(grad_out), ws ? mk(grad_src) : Dst{nullptr, 0, 0},
                     ws ? 1 : 0, wf ? mk(grad_flow) : Dst{nullptr, 0, 0}, wf ? 1 : 0, h, w);
  return dvc::check_launch("flow_warp_backward_kernel");
}

extern "C" int dvc_avg_pool2_backward(const DvcSrc* grad_out, const DvcDst* grad_in, int32_t batch, int32_t h, int32_t w,
                                      float scale, void* stream) {
  DVC_REQUIRE(batch >= 1 && h >= 2 && w >= 2, "dvc_avg_pool2_backward: bad dims");
  CHECK_SRC(grad_out, (long long)(h / 2) * (w / 2), "dvc_avg_pool2_backward");
  CHECK_DST(grad_in, (long long)h * w, "dvc_avg_pool2_backward");
  LAUNCH_1D(avg_pool2_backward_kernel, (long long)batch * grad_out->channels * h * w, stream, mk(grad_out), mk(grad_in),
            batch, h, w, scale);
  return dvc::check_launch("avg_pool2_backward_kernel");
}

extern "C" int dvc_max_pool2_backward(const DvcSrc* x, const DvcSrc* grad_out, const DvcDst* grad_in, int32_t batch,
                                      int32_t h, int32_t w, void* stream) {
  DVC_REQUIRE(batch >= 1 && h >= 2 && w >= 2, "dvc_max_pool2_backward: bad dims");
  CHECK_SRC(x, (long long)h * w, "dvc_max_pool2_backward(x)");
  CHECK_SRC(grad_out, (long long)(h / 2) * (w / 2), "dvc_max_pool2_backward(grad_out)");
  DVC_REQUIRE(x->channels == grad_out->channels, "dvc_max_pool2_backward: channel mismatch");
  CHECK_DST(grad_in, (long long)h * w, "dvc_max_pool2_backward");
  LAUNCH_1D(max_pool2_backward_kernel, (long long)batch * x->channels * h * w, stream, mk(x), mk(grad_out),
            mk(grad_in), batch, h, w);
  return dvc::check_launch("max_pool2_backward_kernel");
}

extern "C" int dvc_bilinear_up2_backward(const DvcSrc* grad_out, const DvcDst* grad_in, int32_t batch, int32_t h,
                                         int32_t w, float scale, void* stream) {
  DVC_REQUIRE(batch >= 1 && h >= 1 && w >= 1, "dvc_bilinear_up2_backward: bad dims");
  CHECK_SRC(grad_out, (long long)4 * h * w, "dvc_bilinear_up2_backward");
  CHECK_DST(grad_in, (long long)h * w, "dvc_bilinear_up2_backward");
  LAUNCH_1D(bilinear_up2_backward_kernel, (long long)batch * grad_out->channels * h * w, stream, mk(grad_out),
            mk(grad_in), batch, h, w, scale);
  return dvc::check_launch("bilinear_up2_backward_kernel");
}

extern "C" int dvc_likelihood_bits_backward(const DvcSrc* lik, const float* grad_bits, const DvcDst* grad_lik,
                                            int32_t batch, int32_t h, int32_t w, void* stream) {
  DVC_REQUIRE(batch >= 1 && h >= 1 && w >= 1 && grad_bits != nullptr, "dvc_likelihood_bits_backward: bad arguments");
  const long long plane = (long long)h * w;
  CHECK_SRC(lik, plane, "dvc_likelihood_bits_backward");
  CHECK_DST(grad_lik, plane, "dvc_likelihood_bits_backward");
  LAUNCH_1D(likelihood_bits_backward_kernel, (long long)batch * lik->channels * plane, stream, mk(lik), grad_bits,
            mk(grad_lik), batch, h, w);
  return dvc::check_launch("likelihood_bits_backward_kernel");
}

extern "C" int dvc_sq_err_backward(const DvcSrc* a, const DvcSrc* b, const float* grad_sse, const DvcDst* grad_a,
                                   int32_t batch, int32_t h, int32_t w, void* stream) {
  DVC_REQUIRE(batch >= 1 && h >= 1 && w >= 1 && grad_sse != nullptr, "dvc_sq_err_backward: bad arguments");
  const long long plane = (long long)h * w;
  CHECK_SRC(a, plane, "dvc_sq_err_backward(a)");
  CHECK_SRC(b, plane, "dvc_sq_err_backward(b)");
  DVC_REQUIRE(a->channels == b->channels, "dvc_sq_err_backward: channel mismatch");
  CHECK_DST(grad_a, plane, "dvc_sq_err_backward");
  LAUNCH_1D(sq_err_backward_kernel, (long long)batch * a->channels * plane, stream, mk(a), mk(b), grad_sse, mk(grad_a),
            batch, h, w);
  return dvc::check_launch("sq_err_backward_kernel");
}
