// Direct (implicit-GEMM) 2-D convolution on the gfx950 matrix cores, exact fp32.
//
// Mapping.  One MFMA v_mfma_f32_32x32x2_f32 multiplies A[32 x 2] by B[2 x 32]:
//   rows  (A's i)  = 32 output channels,
//   cols  (B's j)  = 32 horizontally adjacent output pixels (one per lane & 31),
//   k              = 2 input channels at one filter tap (lane >> 5 picks which).
// So the "im2col" matrix is never built: a lane's B operand for tap (ky,kx) is a
// plain LDS read of the staged input halo tile at a compile-time offset, and the
// A operand is one weight read from an LDS image laid out [k][cout].
// A workgroup (4 waves) owns a (4*NT) x 32 pixel tile times 32*MT output
// channels; a wave owns NT rows of it and keeps MT*NT 32x32 accumulators.
// The result is bit-for-bit a k-ordered fmaf chain (ci-pair major, tap minor,
// lower channel of the pair first).
//
// Fused around the MFMA loop: channel concat of up to 4 sources and LeakyReLU on
// the input while staging; bias, LeakyReLU/ReLU, scaled residual add and the
// PixelShuffle(2) scatter in the epilogue.
#include "dvc_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct ConvArgs {
  DvcSrc src[DVC_MAX_SRC];
  int n_src;
  int cin, h, w;         // input
  int cout, ho, wo;      // conv output (before shuffle)
  const float* weight;
  const float* bias;
  float* out;
  long long obs, ocs;
  const float* res;
  long long rbs, rcs;
  const float* res_scale;
  const float* res2;
  long long r2bs, r2cs;
  int c_final;           // channels of the stored tensor (cout or cout/4)
  float in_slope, out_slope;
  int in_act, out_act, ps;
  int tiles_x;
};

template <int KS, int S>
struct ChunkOf {  // input channels staged per K-chunk (even)
  static constexpr int value = (KS == 7) ? 2 : (KS == 1) ? 16 : (S == 2 ? 4 : 8);
};

template <int KS, int S, int MT, int NT>
__global__ __launch_bounds__(256, 2) void conv_mfma_f32_kernel(const ConvArgs a) {
  constexpr int CK = ChunkOf<KS, S>::value;
  constexpr int TH = 4 * NT, TW = 32, CB = 32 * MT, PAD = KS / 2;
  constexpr int SL = (KS == 1) ? 1 : S;  // pixel stride inside the LDS tile
  constexpr int IH = (TH - 1) * SL + KS;
  constexpr int IW = (TW - 1) * SL + KS;
  constexpr int IWP = IW;
  constexpr int CHS = IH * IWP;          // floats per staged channel
  constexpr int TAPS = KS * KS;
  constexpr int KCH = CK * TAPS;         // k values per chunk
  constexpr int CBP = CB + 1;            // padded cout pitch of the weight image

  __shared__ float in_s[CK * CHS];
  __shared__ float w_s[KCH * CBP];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6, half = lane >> 5, l32 = lane & 31;
  const int tile = blockIdx.x;
  const int ty = tile / a.tiles_x, tx = tile - ty * a.tiles_x;
  const int oy0 = ty * TH, ox0 = tx * TW;
  const int cb0 = blockIdx.y * CB;
  const int n = blockIdx.z;

  f32x16 acc[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

  const float* bbase = in_s + half * CHS + (wave * NT * SL) * IWP + l32 * SL;
  const float* abase = w_s + (half * TAPS) * CBP + l32;

  // top-left input coordinate of the staged tile
  const int gy0 = (KS == 1) ? oy0 * S : oy0 * S - PAD;
  const int gx0 = (KS == 1) ? ox0 * S : ox0 * S - PAD;
  constexpr int GS = (KS == 1) ? S : 1;  // global step between staged pixels

  for (int c0 = 0; c0 < a.cin; c0 += CK) {
    __syncthreads();
    // ---- stage CK input channels (zero outside the image / past Cin)
    for (int c = 0; c < CK; ++c) {
      const int ci = c0 + c;
      const float* plane = nullptr;
      if (ci < a.cin) {
        int cc = ci;
        for (int s = 0; s < a.n_src; ++s) {
          if (cc < a.src[s].channels) {
            plane = a.src[s].ptr + (long long)n * a.src[s].batch_stride +
                    (long long)cc * a.src[s].chan_stride;
            break;
          }
          cc -= a.src[s].channels;
        }
      }
      float* dst = in_s + c * CHS;
      for (int i = tid; i < IH * IW; i += 256) {
        const int iy = i / IW, ix = i - iy * IW;
        const int gy = gy0 + iy * GS, gx = gx0 + ix * GS;
        float v = 0.f;
        if (plane != nullptr && gy >= 0 && gy < a.h && gx >= 0 && gx < a.w)
          v = plane[(long long)gy * a.w + gx];
        if (a.in_act) v = dvc::leaky(v, a.in_slope);
        dst[iy * IWP + ix] = v;
      }
    }
    // ---- stage the weight slab [k][cout] for this chunk (OIHW -> transposed)
    for (int i = tid; i < CB * KCH; i += 256) {
      const int co = i / KCH, kk = i - co * KCH;
      const int ci = c0 + kk / TAPS;
      const int gco = cb0 + co;
      float v = 0.f;
      if (gco < a.cout && ci < a.cin)
        v = a.weight[((long long)gco * a.cin + c0) * TAPS + kk];
      w_s[kk * CBP + co] = v;
    }
    __syncthreads();
    // ---- MFMA over the chunk
#pragma unroll
    for (int cp = 0; cp < CK / 2; ++cp) {
#pragma unroll
      for (int ky = 0; ky < KS; ++ky) {
#pragma unroll
        for (int kx = 0; kx < KS; ++kx) {
          float av[MT], bv[NT];
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
            av[mt] = abase[(2 * cp * TAPS + ky * KS + kx) * CBP + mt * 32];
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
            bv[nt] = bbase[2 * cp * CHS + (nt * SL + ky) * IWP + kx];
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
              acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[mt], bv[nt], acc[mt][nt], 0, 0, 0);
        }
      }
    }
  }

  // ---- epilogue: bias, activation, residual, (pixel-shuffle) store
  const int ox = ox0 + l32;
  if (ox >= a.wo) return;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int oy = oy0 + wave * NT + nt;
      if (oy >= a.ho) continue;
      if (!a.ps) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int co = cb0 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
          if (co >= a.cout) continue;
          float v = acc[mt][nt][r];
          if (a.bias) v += a.bias[co];
          if (a.out_act) v = dvc::leaky(v, a.out_slope);
          const long long pix = (long long)oy * a.wo + ox;
          if (a.res) {
            float rv = a.res[(long long)n * a.rbs + (long long)co * a.rcs + pix];
            if (a.res_scale) rv *= a.res_scale[n * a.c_final + co];
            v += rv;
            if (a.res2) v += a.res2[(long long)n * a.r2bs + (long long)co * a.r2cs + pix];
          }
          a.out[(long long)n * a.obs + (long long)co * a.ocs + pix] = v;
        }
      } else {
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          const int co = cb0 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;  // even
          if (co >= a.cout) continue;
          float v0 = acc[mt][nt][r], v1 = acc[mt][nt][r + 1];
          if (a.bias) { v0 += a.bias[co]; v1 += a.bias[co + 1]; }
          if (a.out_act) { v0 = dvc::leaky(v0, a.out_slope); v1 = dvc::leaky(v1, a.out_slope); }
          const int c2 = co >> 2, dy = (co >> 1) & 1;
          const long long pix = (long long)(2 * oy + dy) * (2 * a.wo) + 2 * ox;
          if (a.res) {
            const float* rp = a.res + (long long)n * a.rbs + (long long)c2 * a.rcs + pix;
            float r0 = rp[0], r1 = rp[1];
            if (a.res_scale) { const float sc = a.res_scale[n * a.c_final + c2]; r0 *= sc; r1 *= sc; }
            v0 += r0; v1 += r1;
            if (a.res2) {
              const float* qp = a.res2 + (long long)n * a.r2bs + (long long)c2 * a.r2cs + pix;
              v0 += qp[0]; v1 += qp[1];
            }
          }
          float2* op = reinterpret_cast<float2*>(a.out + (long long)n * a.obs + (long long)c2 * a.ocs + pix);
          *op = make_float2(v0, v1);
        }
      }
    }
  }
}

template <int KS, int S, int MT, int NT>
int launch(const ConvArgs& a, int batch, hipStream_t st) {
  constexpr int TH = 4 * NT, TW = 32, CB = 32 * MT;
  ConvArgs b = a;
  const int tiles_y = (a.ho + TH - 1) / TH;
  b.tiles_x = (a.wo + TW - 1) / TW;
  dim3 grid(b.tiles_x * tiles_y, (a.cout + CB - 1) / CB, batch);
  hipLaunchKernelGGL((conv_mfma_f32_kernel<KS, S, MT, NT>), grid, dim3(256), 0, st, b);
  return dvc::check_launch("conv_mfma_f32_kernel");
}

template <int KS, int S>
int dispatch_tile(const ConvArgs& a, int batch, hipStream_t st) {
  const int mt = a.cout > 32 ? 2 : 1;
  // big tiles (16 x 32 pixels) only when they still give the 256 CUs several blocks each
  const long long big_blocks = (long long)((a.ho + 15) / 16) * ((a.wo + 31) / 32) *
                               ((a.cout + 32 * mt - 1) / (32 * mt)) * batch;
  const int nt = big_blocks >= 1024 ? 4 : 1;
  if (mt == 2) return nt == 4 ? launch<KS, S, 2, 4>(a, batch, st) : launch<KS, S, 2, 1>(a, batch, st);
  return nt == 4 ? launch<KS, S, 1, 4>(a, batch, st) : launch<KS, S, 1, 1>(a, batch, st);
}

}  // namespace

extern "C" int dvc_conv2d(const DvcConv2d* p, void* stream) {
  DVC_REQUIRE(p != nullptr, "dvc_conv2d: null params");
  DVC_REQUIRE(p->n_src >= 1 && p->n_src <= DVC_MAX_SRC, "dvc_conv2d: n_src=%d not in 1..%d", p->n_src, DVC_MAX_SRC);
  DVC_REQUIRE(p->ksize == 1 || p->ksize == 3 || p->ksize == 7, "dvc_conv2d: ksize=%d unsupported", p->ksize);
  DVC_REQUIRE(p->stride == 1 || p->stride == 2, "dvc_conv2d: stride=%d unsupported", p->stride);
  DVC_REQUIRE(!(p->ksize == 7 && p->stride != 1), "dvc_conv2d: 7x7 stride 2 unsupported");
  DVC_REQUIRE(p->batch >= 1 && p->in_h >= 1 && p->in_w >= 1 && p->out_channels >= 1, "dvc_conv2d: bad dims");
  DVC_REQUIRE(p->batch <= 65535, "dvc_conv2d: batch too large");
  DVC_REQUIRE(p->weight != nullptr && p->out.ptr != nullptr, "dvc_conv2d: null weight/out");
  DVC_REQUIRE(!p->pixel_shuffle || p->out_channels % 4 == 0, "dvc_conv2d: pixel_shuffle needs Cout %% 4 == 0");
  ConvArgs a{};
  int cin = 0;
  for (int s = 0; s < p->n_src; ++s) {
    DVC_REQUIRE(p->src[s].ptr != nullptr && p->src[s].channels >= 1, "dvc_conv2d: bad source %d", s);
    DVC_REQUIRE(p->src[s].chan_stride >= (int64_t)p->in_h * p->in_w, "dvc_conv2d: source %d channel stride < H*W", s);
    a.src[s] = p->src[s];
    cin += p->src[s].channels;
  }
  a.n_src = p->n_src;
  a.cin = cin;
  a.h = p->in_h;
  a.w = p->in_w;
  a.cout = p->out_channels;
  const int pad = p->ksize / 2;
  a.ho = (p->in_h + 2 * pad - p->ksize) / p->stride + 1;
  a.wo = (p->in_w + 2 * pad - p->ksize) / p->stride + 1;
  a.weight = p->weight;
  a.bias = p->bias;
  a.out = p->out.ptr;
  a.obs = p->out.batch_stride;
  a.ocs = p->out.chan_stride;
  a.res = p->residual;
  a.rbs = p->res_batch_stride;
  a.rcs = p->res_chan_stride;
  a.res_scale = p->res_scale;
  a.res2 = p->residual2;
  a.r2bs = p->res2_batch_stride;
  a.r2cs = p->res2_chan_stride;
  a.ps = p->pixel_shuffle ? 1 : 0;
  a.c_final = a.ps ? a.cout / 4 : a.cout;
  const long long plane = (long long)a.ho * a.wo * (a.ps ? 4 : 1);
  DVC_REQUIRE(a.ocs >= plane, "dvc_conv2d: out channel stride %lld < plane %lld", a.ocs, plane);
  DVC_REQUIRE(!a.res || a.rcs >= plane, "dvc_conv2d: residual channel stride too small");
  DVC_REQUIRE(!a.res2 || (a.res && a.r2cs >= plane), "dvc_conv2d: residual2 needs residual and a full channel stride");
  DVC_REQUIRE(!a.ps || (reinterpret_cast<uintptr_t>(a.out) % 8 == 0 && a.obs % 2 == 0 && a.ocs % 2 == 0),
              "dvc_conv2d: pixel_shuffle output must be 8-byte aligned with even strides");
  DVC_REQUIRE((long long)((a.ho + 3) / 4) * ((a.wo + 31) / 32) < 2147483647LL, "dvc_conv2d: too many tiles");
  a.in_slope = p->in_slope;
  a.out_slope = p->out_slope;
  a.in_act = p->in_act ? 1 : 0;
  a.out_act = p->out_act ? 1 : 0;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (p->ksize == 3) return p->stride == 1 ? dispatch_tile<3, 1>(a, p->batch, st) : dispatch_tile<3, 2>(a, p->batch, st);
  if (p->ksize == 1) return p->stride == 1 ? dispatch_tile<1, 1>(a, p->batch, st) : dispatch_tile<1, 2>(a, p->batch, st);
  return dispatch_tile<7, 1>(a, p->batch, st);
}
