// Weight gradient of the 2-D convolutions on the gfx950 matrix cores, exact fp32.
//
//   gW[co][ci][ky][kx] = sum over (n, oy, ox) of  g[n][co][oy][ox] * x[n][ci][oy*s + ky - p][ox*s + kx - p]
//
// As a GEMM this is [Cout x pixels] * [pixels x (Cin * taps)]: a small output and a very long
// reduction.  One v_mfma_f32_32x32x2_f32 takes rows = 32 output channels, columns = 32 input
// channels (at one tap), k = 2 horizontally adjacent output pixels (lane >> 5 picks which).
// A workgroup (8 waves) owns a (32 MT) x (32 CT) block (Cout x Cin; MT, CT in {1, 2} by the
// layer's channel counts) of the result for all taps (7x7: one tap row per workgroup, 7
// workgroups in z).  A wave keeps one 32 x 32 corner of the block for every tap in registers
// (9 accumulators for 3x3); the 8 / (MT CT) waves that share a corner take interleaved pixel
// pairs of each tile and their sums are added through LDS, in a fixed order, when the workgroup
// is done.  The pixel axis is cut into slabs: workgroup `split` of a block walks the tiles
// split, split + S, ... and writes one partial block; a second kernel adds the partials in a
// fixed order (deterministic, no atomics).
//
// Staging per pixel tile (TH x 32 output pixels): the matching halo tile of 64 input channels
// and the gradient tile of 64 output channels go global -> registers (range-checked buffer
// loads: padding and channels past the end arrive as zeros) while the previous tile is being
// multiplied, then into LDS laid out [channel][pixel] with an odd channel stride, so that the
// MFMA operands -- 32 lanes = 32 channels at one pixel -- are conflict-free LDS reads.
#include <atomic>

#include "dvc_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct WgradArgs {
  DvcSrc src[DVC_MAX_SRC];
  int n_src;
  int cin, h, w;
  int cout, ho, wo;
  int batch;
  const float* g;     // [N][cout][ho][wo], contiguous
  float* partial;     // [S][taps][cout_pad][cin_pad]
  int cout_pad, cin_pad, n_cib;
  int tiles_x, tiles_y, n_tiles;
  float in_slope;     // LeakyReLU applied to x while staging (1 = none)
};

template <int KS, int S, int MT, int CT>
struct WGeo {
  static constexpr int KG = 8 / (MT * CT);      // waves sharing one corner (pixel-pair interleave)
  static constexpr int XCH = 32 * CT, GCH = 32 * MT;   // staged input / gradient channels
  static constexpr int KH = KS == 7 ? 1 : KS;   // tap rows per workgroup
  static constexpr int TAPS = KH * KS;
  static constexpr int TAPS_ALL = KS * KS;
  static constexpr int TH = (S == 2 && KS == 3) ? 1 : 2, TW = 32, PAD = KS / 2;
  static constexpr int SL = (KS == 1) ? 1 : S;   // pixel stride inside the LDS tile
  static constexpr int GS = (KS == 1) ? S : 1;   // global step between staged pixels
  static constexpr int IH = (TH - 1) * SL + KH, IW = (TW - 1) * SL + KS;
  static constexpr int XE = IH * IW, XSTR = XE | 1, XSLOTS = (XE + 63) / 64;
  static constexpr int GE = TH * TW, GSTR = GE + 1;
  static constexpr int STAGE_FLOATS = XCH * XSTR + GCH * GSTR;
  static constexpr int RED_FLOATS = (KG - 1) * MT * CT * 1024;   // one tap's tiles of the other groups
  static constexpr int LDS_BYTES = (STAGE_FLOATS > RED_FLOATS ? STAGE_FLOATS : RED_FLOATS) * 4;
};

template <int KS, int S, int MT, int CT>
__global__ __launch_bounds__(512) void conv_wgrad_kernel(const WgradArgs a) {
  using G = WGeo<KS, S, MT, CT>;
  constexpr int KG = G::KG, XCH = G::XCH, GCH = G::GCH, XPW = XCH / 8, GPW = GCH / 8;   // channels staged per wave
  constexpr int TAPS = G::TAPS, TH = G::TH, SL = G::SL, GS = G::GS, IW = G::IW, XE = G::XE, XSTR = G::XSTR;
  constexpr int XSLOTS = G::XSLOTS, GE = G::GE, GSTR = G::GSTR, PAD = G::PAD, KH = G::KH;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* const x_s = lds;                  // [XCH][XSTR]
  float* const g_s = lds + XCH * XSTR;     // [GCH][GSTR]

  const int tid = threadIdx.x;
  const int lane = tid & 63, half = lane >> 5, l32 = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int corner = wave % (MT * CT);        // which 32 x 32 corner of the block
  const int grp = wave / (MT * CT);           // which interleave of the tile's pixel pairs
  const int mt = corner / CT, ct = corner % CT;
  const int cob = blockIdx.y / a.n_cib, cib = blockIdx.y - cob * a.n_cib;
  const int co0 = cob * GCH, ci0 = cib * XCH;
  const int ky0 = (KS == 7) ? (int)blockIdx.z : 0;
  const int nsplit = gridDim.x;

  // concatenated sources as scalars (see the forward kernel: select chains become private tables)
  const unsigned long long b0 = reinterpret_cast<unsigned long long>(a.src[0].ptr);
  const unsigned long long b1 = reinterpret_cast<unsigned long long>(a.src[1].ptr);
  const unsigned long long b2 = reinterpret_cast<unsigned long long>(a.src[2].ptr);
  const unsigned long long b3 = reinterpret_cast<unsigned long long>(a.src[3].ptr);
  const long long cs0 = a.src[0].chan_stride, cs1 = a.src[1].chan_stride;
  const long long cs2 = a.src[2].chan_stride, cs3 = a.src[3].chan_stride;
  const long long bs0 = a.src[0].batch_stride, bs1 = a.src[1].batch_stride;
  const long long bs2 = a.src[2].batch_stride, bs3 = a.src[3].batch_stride;
  const int e0 = a.src[0].channels, e1 = e0 + a.src[1].channels, e2 = e1 + a.src[2].channels;
  const int cin = a.cin, cout = a.cout;
  const unsigned xplane_bytes = 4u * (unsigned)(a.h * a.w), gplane_bytes = 4u * (unsigned)(a.ho * a.wo);
  const long long gplane = (long long)a.ho * a.wo;
  const float in_slope = a.in_slope;
  const int per_n = a.tiles_x * a.tiles_y;

  float xr[XPW][XSLOTS], gr[GPW];   // wave w stages channels XPW w .. of x and GPW w .. of g

  auto prefetch = [&](int t) {
    const int n = t / per_n, r = t - n * per_n;
    const int ty = r / a.tiles_x, tx = r - ty * a.tiles_x;
    const int oy0 = ty * TH, ox0 = tx * 32;
    const int gy0 = (KS == 1) ? oy0 * S : oy0 * S - PAD + ky0;
    const int gx0 = (KS == 1) ? ox0 * S : ox0 * S - PAD;
    unsigned xoff[XSLOTS];
#pragma unroll
    for (int j = 0; j < XSLOTS; ++j) {
      const int e = lane + 64 * j;
      const int iy = e / IW, ix = e - iy * IW;
      const int gy = gy0 + iy * GS, gx = gx0 + ix * GS;
      const bool ok = e < XE && gy >= 0 && gy < a.h && gx >= 0 && gx < a.w;
      xoff[j] = ok ? 4u * (unsigned)(gy * a.w + gx) : 0x80000000u;
    }
    unsigned goff;
    {
      const int row = lane >> 5, col = lane & 31;
      const int oy = oy0 + row, ox = ox0 + col;
      const bool ok = lane < GE && oy < a.ho && ox < a.wo;
      goff = ok ? 4u * (unsigned)(oy * a.wo + ox) : 0x80000000u;
    }
#pragma unroll
    for (int c = 0; c < XPW; ++c) {
      const int ci = ci0 + XPW * wave + c;
      const unsigned long long m0 = 0ull - (unsigned long long)(ci >= e0);
      const unsigned long long m1 = 0ull - (unsigned long long)(ci >= e1);
      const unsigned long long m2 = 0ull - (unsigned long long)(ci >= e2);
      const unsigned long long base = b0 + ((b1 - b0) & m0) + ((b2 - b1) & m1) + ((b3 - b2) & m2);
      const long long cs = cs0 + ((cs1 - cs0) & (long long)m0) + ((cs2 - cs1) & (long long)m1) + ((cs3 - cs2) & (long long)m2);
      const long long bs = bs0 + ((bs1 - bs0) & (long long)m0) + ((bs2 - bs1) & (long long)m1) + ((bs3 - bs2) & (long long)m2);
      const int first = (e0 & (int)m0) + ((e1 - e0) & (int)m1) + ((e2 - e1) & (int)m2);
      const bool have = ci < cin;
      const unsigned long long pl = have ? base + 4ull * (unsigned long long)((long long)n * bs + (long long)(ci - first) * cs) : b0;
      const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)pl);
      const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(pl >> 32));
      const unsigned nb = __builtin_amdgcn_readfirstlane(have ? xplane_bytes : 0u);
      const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
          reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo), 0, nb, 0x00020000);
#pragma unroll
      for (int j = 0; j < XSLOTS; ++j)
        xr[c][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, xoff[j], 0, 0));
    }
#pragma unroll
    for (int c = 0; c < GPW; ++c) {
      const int co = co0 + GPW * wave + c;
      const bool haveg = co < cout;
      const unsigned long long pg = reinterpret_cast<unsigned long long>(a.g) +
                                    4ull * (unsigned long long)(((long long)n * cout + (haveg ? co : 0)) * gplane);
      const unsigned glo = __builtin_amdgcn_readfirstlane((unsigned)pg);
      const unsigned ghi = __builtin_amdgcn_readfirstlane((unsigned)(pg >> 32));
      const unsigned gnb = __builtin_amdgcn_readfirstlane(haveg ? gplane_bytes : 0u);
      const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc(
          reinterpret_cast<void*>(((unsigned long long)ghi << 32) | glo), 0, gnb, 0x00020000);
      gr[c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rg, goff, 0, 0));
    }
  };

  auto commit = [&]() {
#pragma unroll
    for (int c = 0; c < XPW; ++c) {
      float* xd = x_s + (XPW * wave + c) * XSTR + lane;
#pragma unroll
      for (int j = 0; j < XSLOTS; ++j)
        if (64 * (j + 1) <= XE || lane + 64 * j < XE) xd[64 * j] = fmaxf(xr[c][j], xr[c][j] * in_slope);
    }
#pragma unroll
    for (int c = 0; c < GPW; ++c)
      if (GE == 64 || lane < GE) g_s[(GPW * wave + c) * GSTR + lane] = gr[c];
  };

  f32x16 acc[TAPS];
#pragma unroll
  for (int t = 0; t < TAPS; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  const float* const a_base = g_s + (mt * 32 + l32) * GSTR + half + 2 * grp;
  const float* const b_base = x_s + (ct * 32 + l32) * XSTR + (half + 2 * grp) * SL;
  static_assert(32 % (2 * KG) == 0 && GE % (2 * KG) == 0, "pixel-pair interleave must tile a row");

  int t = blockIdx.x;
  if (t < a.n_tiles) {
    prefetch(t);
    commit();
  }
  __syncthreads();
  for (; t < a.n_tiles; t += nsplit) {
    const int tn = t + nsplit;
    if (tn < a.n_tiles) prefetch(tn);
#pragma unroll
    for (int jj = 0; jj < GE / (2 * KG); ++jj) {   // this wave's pixel pairs: pixels 2 KG jj + 2 grp, + 1
      const int row = (2 * KG * jj) / 32, px = (2 * KG * jj) % 32;
      const float av = a_base[row * 32 + px];
#pragma unroll
      for (int ky = 0; ky < KH; ++ky)
#pragma unroll
        for (int kx = 0; kx < KS; ++kx) {
          const float bv = b_base[(row * SL + ky) * IW + px * SL + kx];
          acc[ky * KS + kx] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[ky * KS + kx], 0, 0, 0);
        }
    }
    __syncthreads();                     // every wave is done reading this tile
    if (tn < a.n_tiles) commit();
    __syncthreads();
  }

  // Sum the KG waves that share a corner: per tap, groups 1 .. KG-1 park their tile in LDS and
  // group 0 adds them in ascending order, then stores the tap of the partial block
  // [split][tap][co][ci]; register r of a tile is row (r&3) + 8 (r>>2) + 4 half, column l32.
  float* const red = lds;   // [KG-1][MT*CT][16][64]; the staging buffers are free after the last barrier
  float* const pbase = a.partial + (long long)blockIdx.x * G::TAPS_ALL * a.cout_pad * a.cin_pad;
#pragma unroll
  for (int tp = 0; tp < TAPS; ++tp) {
    if (grp > 0) {
      float* dst = red + ((grp - 1) * (MT * CT) + corner) * 1024 + lane;
#pragma unroll
      for (int r = 0; r < 16; ++r) dst[64 * r] = acc[tp][r];
    }
    __syncthreads();
    if (grp == 0) {
      f32x16 v = acc[tp];
#pragma unroll
      for (int k = 1; k < KG; ++k) {
        const float* src = red + ((k - 1) * (MT * CT) + corner) * 1024 + lane;
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] += src[64 * r];
      }
      const int tap_all = ky0 * KS + tp;   // 7x7: tp is kx of tap row ky0; otherwise ky0 = 0
      float* const pt = pbase + (long long)tap_all * a.cout_pad * a.cin_pad + ci0 + ct * 32 + l32;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = co0 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        pt[(long long)co * a.cin_pad] = v[r];
      }
    }
    __syncthreads();
  }
}

// gW[co][ci][tap] = sum over splits (ascending) of partial[split][tap][co][ci]
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* partial, float* gw, int nsplit, int taps,
                                                           int cout, int cin, int cout_pad, int cin_pad) {
  const long long total = (long long)cout * taps * cin;
  const long long sstride = (long long)taps * cout_pad * cin_pad;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int ci = (int)(i % cin);
    const long long q = i / cin;
    const int tap = (int)(q % taps), co = (int)(q / taps);
    const float* p = partial + ((long long)tap * cout_pad + co) * cin_pad + ci;
    float s = 0.f;
    for (int k = 0; k < nsplit; ++k) s += p[k * sstride];
    gw[((long long)co * cin + ci) * taps + tap] = s;
  }
}

struct Plan {
  int ho, wo, th, tiles_x, tiles_y, n_tiles, cout_pad, cin_pad, n_cib, blocks, nsplit, taps, mt, ct;
};

int make_plan(int cout, int cin, int ks, int stride, int batch, int h, int w, Plan* p, const char* who) {
  DVC_REQUIRE(ks == 1 || ks == 3 || ks == 7, "%s: ksize=%d unsupported", who, ks);
  DVC_REQUIRE(stride == 1 || stride == 2, "%s: stride=%d unsupported", who, stride);
  DVC_REQUIRE(!(ks == 7 && stride != 1), "%s: 7x7 stride 2 unsupported", who);
  DVC_REQUIRE(cout >= 1 && cin >= 1 && batch >= 1 && h >= 1 && w >= 1, "%s: bad dims", who);
  DVC_REQUIRE((long long)h * w < (1LL << 28), "%s: plane too large", who);
  const int pad = ks / 2;
  p->ho = (h + 2 * pad - ks) / stride + 1;
  p->wo = (w + 2 * pad - ks) / stride + 1;
  p->th = (stride == 2 && ks == 3) ? 1 : 2;
  p->tiles_x = (p->wo + 31) / 32;
  p->tiles_y = (p->ho + p->th - 1) / p->th;
  const long long nt = (long long)batch * p->tiles_x * p->tiles_y;
  DVC_REQUIRE(nt < (1LL << 30), "%s: too many tiles", who);
  p->n_tiles = (int)nt;
  p->mt = cout > 32 ? 2 : 1;
  p->ct = cin > 32 ? 2 : 1;
  const int cob = 32 * p->mt, cib = 32 * p->ct;
  p->cout_pad = (cout + cob - 1) / cob * cob;
  p->cin_pad = (cin + cib - 1) / cib * cib;
  p->n_cib = p->cin_pad / cib;
  p->blocks = (p->cout_pad / cob) * p->n_cib;
  p->taps = ks * ks;
  const int wg_per_split = p->blocks * (ks == 7 ? 7 : 1);
  int s = (256 + wg_per_split - 1) / wg_per_split;   // one 8-wave workgroup fits a CU: about one round of them
  if (s > p->n_tiles) s = p->n_tiles;
  if (s < 1) s = 1;
  p->nsplit = s;
  return DVC_OK;
}

template <int KS, int S, int MT, int CT>
int launch(const WgradArgs& a, const Plan& p, hipStream_t st) {
  using G = WGeo<KS, S, MT, CT>;
  // LDS above the 64 KiB default needs an opt-in per kernel and per device; calls may come from
  // several host threads, so the done-mask is atomic (setting the attribute twice is harmless)
  static std::atomic<unsigned long long> attr_done{0};
  int device = 0;
  if (hipGetDevice(&device) != hipSuccess || device < 0 || device > 63) device = 0;
  const unsigned long long device_bit = 1ull << device;
  if (!(attr_done.load(std::memory_order_acquire) & device_bit)) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_kernel<KS, S, MT, CT>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES);
    if (e != hipSuccess) {
      dvc::set_error("hipFuncSetAttribute(wgrad %d,%d): %s", KS, S, hipGetErrorString(e));
      return DVC_EHIP;
    }
    attr_done.fetch_or(device_bit, std::memory_order_release);
  }
  dim3 grid(p.nsplit, p.blocks, KS == 7 ? 7 : 1);
  hipLaunchKernelGGL((conv_wgrad_kernel<KS, S, MT, CT>), grid, dim3(512), G::LDS_BYTES, st, a);
  return dvc::check_launch("conv_wgrad_kernel");
}

template <int KS, int S>
int launch_block(const WgradArgs& a, const Plan& p, hipStream_t st) {
  if (p.mt == 2) return p.ct == 2 ? launch<KS, S, 2, 2>(a, p, st) : launch<KS, S, 2, 1>(a, p, st);
  return p.ct == 2 ? launch<KS, S, 1, 2>(a, p, st) : launch<KS, S, 1, 1>(a, p, st);
}

}  // namespace

extern "C" int64_t dvc_conv2d_wgrad_scratch_floats(int32_t cout, int32_t cin, int32_t ksize, int32_t stride,
                                                   int32_t batch, int32_t in_h, int32_t in_w) {
  Plan p;
  if (make_plan(cout, cin, ksize, stride, batch, in_h, in_w, &p, "dvc_conv2d_wgrad_scratch_floats")) return DVC_EINVAL;
  return (int64_t)p.nsplit * p.taps * p.cout_pad * p.cin_pad;
}

extern "C" int dvc_conv2d_wgrad(const DvcSrc* srcs, int32_t n_src, int32_t batch, int32_t in_h, int32_t in_w,
                                int32_t in_act, float in_slope, const float* grad_out, int32_t out_channels,
                                int32_t ksize, int32_t stride, float* scratch, float* grad_weight, void* stream) {
  DVC_REQUIRE(srcs != nullptr && n_src >= 1 && n_src <= DVC_MAX_SRC, "dvc_conv2d_wgrad: n_src=%d not in 1..%d", n_src,
              DVC_MAX_SRC);
  DVC_REQUIRE(grad_out != nullptr && scratch != nullptr && grad_weight != nullptr, "dvc_conv2d_wgrad: null pointer");
  DVC_REQUIRE(!in_act || (in_slope >= 0.f && in_slope <= 1.f), "dvc_conv2d_wgrad: in_slope %g outside [0, 1]",
              (double)in_slope);
  WgradArgs a{};
  int cin = 0;
  for (int s = 0; s < n_src; ++s) {
    DVC_REQUIRE(srcs[s].ptr != nullptr && srcs[s].channels >= 1, "dvc_conv2d_wgrad: bad source %d", s);
    DVC_REQUIRE(srcs[s].chan_stride >= (int64_t)in_h * in_w, "dvc_conv2d_wgrad: source %d channel stride < H*W", s);
    a.src[s] = srcs[s];
    cin += srcs[s].channels;
  }
  Plan p;
  if (int rc = make_plan(out_channels, cin, ksize, stride, batch, in_h, in_w, &p, "dvc_conv2d_wgrad")) return rc;
  DVC_REQUIRE((long long)p.ho * p.wo * out_channels * 4 < (1LL << 40), "dvc_conv2d_wgrad: gradient too large");
  a.n_src = n_src;
  a.cin = cin;
  a.h = in_h;
  a.w = in_w;
  a.cout = out_channels;
  a.ho = p.ho;
  a.wo = p.wo;
  a.batch = batch;
  a.g = grad_out;
  a.partial = scratch;
  a.cout_pad = p.cout_pad;
  a.cin_pad = p.cin_pad;
  a.n_cib = p.n_cib;
  a.tiles_x = p.tiles_x;
  a.tiles_y = p.tiles_y;
  a.n_tiles = p.n_tiles;
  a.in_slope = in_act ? in_slope : 1.0f;
  hipStream_t st = static_cast<hipStream_t>(stream);
  int rc;
  if (ksize == 3) rc = stride == 1 ? launch_block<3, 1>(a, p, st) : launch_block<3, 2>(a, p, st);
  else if (ksize == 1) rc = stride == 1 ? launch_block<1, 1>(a, p, st) : launch_block<1, 2>(a, p, st);
  else rc = launch_block<7, 1>(a, p, st);
  if (rc) return rc;
  const long long total = (long long)out_channels * cin * p.taps;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(dvc::stream_grid(total, 256)), dim3(256), 0, st, scratch, grad_weight,
                     p.nsplit, p.taps, out_channels, cin, p.cout_pad, p.cin_pad);
  return dvc::check_launch("wgrad_reduce_kernel");
}
