// Weight gradient of the 2-D convolutions on the gfx950 matrix cores, exact fp32.
//
//   gW[co][ci][ky][kx] = sum over (n, oy, ox) of  g[n][co][oy][ox] * x[n][ci][oy*s + ky - p][ox*s + kx - p]
//
// As a GEMM this is [Cout x pixels] * [pixels x (Cin * taps)]: a small output and a very long
// reduction.  One v_mfma_f32_32x32x2_f32 takes rows = 32 output channels, columns = 32 input
// channels (at one tap), k = 2 horizontally adjacent output pixels (lane >> 5 picks which).
// A workgroup (8 waves) owns a (32 MT) x (32 CT) block (Cout x Cin; MT, CT in {1, 2} by the
// layer's channel counts) of the result for all taps (7x7 has a kernel of its own, below).  A wave keeps one 32 x 32 corner of the block for every tap in registers
// (9 accumulators for 3x3); the 8 / (MT CT) waves that share a corner take interleaved pixel
// pairs of each tile and their sums are added through LDS, in a fixed order, when the workgroup
// is done.  The pixel axis is cut into slabs: workgroup `split` of a block walks the tiles
// split, split + S, ... and writes one partial block; a second kernel adds the partials in a
// fixed order (deterministic, no atomics).  The bias gradient (sum of g over pixels) rides along: the waves
// that stage the gradient tiles add up what they stage.
//
// Staging per pixel tile (TH x 32 output pixels): the matching halo tile of 64 input channels
// and the gradient tile of 64 output channels go global -> registers (range-checked buffer
// loads: padding and channels past the end arrive as zeros) while the previous tile is being
// multiplied, then into LDS laid out [channel][pixel] with an odd channel stride, so that the
// MFMA operands -- 32 lanes = 32 channels at one pixel -- are conflict-free LDS reads.
#include <atomic>

#include "dvc_common.h"

#ifndef WG_TH_S1
#define WG_TH_S1 4
#endif
#ifndef WG_TH_S2
#define WG_TH_S2 2
#endif

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
  float* bias_partial;   // [S][cout_pad] per-split sums of g over pixels (bias gradient), or null
  int cout_pad, cin_pad, n_cib;
  int tiles_x, tiles_y, n_tiles;
  float in_slope;     // LeakyReLU applied to x while staging (1 = none)
};


__device__ __forceinline__ unsigned long long uniform64(unsigned long long v) {
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v);
  const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return ((unsigned long long)hi << 32) | lo;
}

// Where channel `ci` of the concatenated sources lives: address of its plane in sample 0, byte
// stride between samples, bytes of the plane (0 when ci is past the last channel, so that every
// range-checked load of it returns zero).  Mask arithmetic instead of select chains: those become
// private lookup tables.  All results are wave-uniform (ci must be).
__device__ __forceinline__ void source_plane(const WgradArgs& a, int ci, unsigned plane_bytes, unsigned long long* p0,
                                             unsigned long long* sample_bytes, unsigned* nbytes) {
  const unsigned long long b0 = reinterpret_cast<unsigned long long>(a.src[0].ptr);
  const unsigned long long b1 = reinterpret_cast<unsigned long long>(a.src[1].ptr);
  const unsigned long long b2 = reinterpret_cast<unsigned long long>(a.src[2].ptr);
  const unsigned long long b3 = reinterpret_cast<unsigned long long>(a.src[3].ptr);
  const long long cs0 = a.src[0].chan_stride, cs1 = a.src[1].chan_stride;
  const long long cs2 = a.src[2].chan_stride, cs3 = a.src[3].chan_stride;
  const long long bs0 = a.src[0].batch_stride, bs1 = a.src[1].batch_stride;
  const long long bs2 = a.src[2].batch_stride, bs3 = a.src[3].batch_stride;
  const int e0 = a.src[0].channels, e1 = e0 + a.src[1].channels, e2 = e1 + a.src[2].channels;
  const unsigned long long m0 = 0ull - (unsigned long long)(ci >= e0);
  const unsigned long long m1 = 0ull - (unsigned long long)(ci >= e1);
  const unsigned long long m2 = 0ull - (unsigned long long)(ci >= e2);
  const unsigned long long base = b0 + ((b1 - b0) & m0) + ((b2 - b1) & m1) + ((b3 - b2) & m2);
  const long long cs = cs0 + ((cs1 - cs0) & (long long)m0) + ((cs2 - cs1) & (long long)m1) + ((cs3 - cs2) & (long long)m2);
  const long long bs = bs0 + ((bs1 - bs0) & (long long)m0) + ((bs2 - bs1) & (long long)m1) + ((bs3 - bs2) & (long long)m2);
  const int first = (e0 & (int)m0) + ((e1 - e0) & (int)m1) + ((e2 - e1) & (int)m2);
  const bool have = ci < a.cin;
  *p0 = uniform64(have ? base + 4ull * (unsigned long long)((long long)(ci - first) * cs) : b0);
  *sample_bytes = uniform64(have ? 4ull * (unsigned long long)bs : 0ull);
  *nbytes = __builtin_amdgcn_readfirstlane(have ? plane_bytes : 0u);
}

__device__ __forceinline__ __amdgpu_buffer_rsrc_t plane_rsrc(unsigned long long p0, unsigned long long sample_bytes, int n,
                                                             unsigned nbytes) {
  const unsigned long long pl = p0 + (unsigned long long)n * sample_bytes;
  return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(pl), 0, nbytes, 0x00020000);
}

// Between the MFMA groups of a tile the loads of the next tile are issued one channel at a time:
// scalar address arithmetic and load issue run while the matrix pipe works.  The scheduler must not
// gather those pieces in one place; only LDS reads may move across (0x100).
#define WG_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0x100)

// output rows per staged pixel tile (x 32 columns)
constexpr int tile_rows(int ks, int s) { return ks == 7 ? 4 : (s == 2 && ks == 3) ? WG_TH_S2 : WG_TH_S1; }

// TW: output pixels per tile row.  32, or 16 (with twice the rows) for planes at most 16 pixels wide -- the latent
// layers of a 256 x 256 training crop are 16 x 16, and half of every 32-pixel row was padding.
template <int KS, int S, int MT, int CT, int TW_ = 32>
struct WGeo {
  static constexpr int KG = 8 / (MT * CT);      // waves sharing one corner (pixel-pair interleave)
  static constexpr int XCH = 32 * CT, GCH = 32 * MT;   // staged input / gradient channels
  static_assert(KS == 1 || KS == 3, "7x7 has its own kernel");
  static constexpr int KH = KS;
  static constexpr int TAPS = KS * KS;
  static_assert(TW_ == 32 || TW_ == 16, "tile rows of 32 or 16 pixels");
  static constexpr int TW = TW_, TH = tile_rows(KS, S) * (32 / TW_), PAD = KS / 2;
  static constexpr int SL = (KS == 1) ? 1 : S;   // pixel stride inside the LDS tile
  static constexpr int GS = (KS == 1) ? S : 1;   // global step between staged pixels
  static constexpr int IH = (TH - 1) * SL + KH, IW = (TW - 1) * SL + KS;
  static constexpr int XE = IH * IW, XSTR = XE | 1, XSLOTS = (XE + 63) / 64;
  static constexpr int GE = TH * TW, GSTR = GE + 1, GSLOTS = (GE + 63) / 64;
  static constexpr int STAGE_FLOATS = XCH * XSTR + GCH * GSTR;
  static constexpr int RED_FLOATS = (KG - 1) * MT * CT * 1024;   // one tap's tiles of the other groups
  static constexpr int LDS_BYTES = (STAGE_FLOATS > RED_FLOATS ? STAGE_FLOATS : RED_FLOATS) * 4;
};

template <int KS, int S, int MT, int CT, int TW = 32>
__global__ __launch_bounds__(512) void conv_wgrad_kernel(const WgradArgs a) {
  using G = WGeo<KS, S, MT, CT, TW>;
  constexpr int KG = G::KG, XCH = G::XCH, GCH = G::GCH, XPW = XCH / 8, GPW = GCH / 8;   // channels staged per wave
  constexpr int TAPS = G::TAPS, TH = G::TH, SL = G::SL, GS = G::GS, IW = G::IW, XE = G::XE, XSTR = G::XSTR;
  constexpr int XSLOTS = G::XSLOTS, GE = G::GE, GSTR = G::GSTR, GSLOTS = G::GSLOTS, PAD = G::PAD, KH = G::KH;
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
  const int nsplit = gridDim.x;

  const int cout = a.cout;
  const unsigned xplane_bytes = 4u * (unsigned)(a.h * a.w), gplane_bytes = 4u * (unsigned)(a.ho * a.wo);
  const float in_slope = a.in_slope;
  const int per_n = a.tiles_x * a.tiles_y;

  // wave w stages channels XPW w .. of x and GPW w .. of g; their planes, once
  unsigned long long xp[XPW], xsb[XPW], gp[GPW];
  unsigned xnb[XPW], gnb[GPW];
#pragma unroll
  for (int c = 0; c < XPW; ++c) source_plane(a, ci0 + XPW * wave + c, xplane_bytes, &xp[c], &xsb[c], &xnb[c]);
  const unsigned long long gsb = (unsigned long long)cout * (unsigned long long)gplane_bytes;
#pragma unroll
  for (int c = 0; c < GPW; ++c) {
    const int co = co0 + GPW * wave + c;
    const bool haveg = co < cout;
    gp[c] = uniform64(reinterpret_cast<unsigned long long>(a.g) + (haveg ? (unsigned long long)co * gplane_bytes : 0ull));
    gnb[c] = __builtin_amdgcn_readfirstlane(haveg ? gplane_bytes : 0u);
  }

  float xr[XPW][XSLOTS], gr[GPW][GSLOTS];
  float gsum[GPW];                  // per lane: sum of the gradient values this lane staged (bias gradient)
#pragma unroll
  for (int c = 0; c < GPW; ++c) gsum[c] = 0.f;

  // per-lane offsets of tile t inside a plane (invalid offsets fail the range check and load zeros)
  unsigned xoff[XSLOTS], goff[GSLOTS];
  int pf_n = 0;
  unsigned pf_live = 0;   // all-ones when the tile exists
  auto prefetch_setup = [&](int t) {
    pf_live = t < a.n_tiles ? 0xffffffffu : 0u;
    const int n = t / per_n, r = t - n * per_n;
    const int ty = r / a.tiles_x, tx = r - ty * a.tiles_x;
    const int oy0 = ty * TH, ox0 = tx * TW;
    const int gy0 = (KS == 1) ? oy0 * S : oy0 * S - PAD;
    const int gx0 = (KS == 1) ? ox0 * S : ox0 * S - PAD;
    pf_n = n;
#pragma unroll
    for (int j = 0; j < XSLOTS; ++j) {
      const int e = lane + 64 * j;
      const int iy = e / IW, ix = e - iy * IW;
      const int gy = gy0 + iy * GS, gx = gx0 + ix * GS;
      const bool ok = e < XE && gy >= 0 && gy < a.h && gx >= 0 && gx < a.w;
      xoff[j] = ok ? 4u * (unsigned)(gy * a.w + gx) : 0x80000000u;
    }
#pragma unroll
    for (int j = 0; j < GSLOTS; ++j) {
      const int row = (lane + 64 * j) / TW, col = (lane + 64 * j) % TW;
      const int oy = oy0 + row, ox = ox0 + col;
      const bool ok = lane + 64 * j < GE && oy < a.ho && ox < a.wo;
      goff[j] = ok ? 4u * (unsigned)(oy * a.wo + ox) : 0x80000000u;
    }
  };
  // piece k < XPW: the loads of x channel k; XPW <= k < XPW + GPW: those of g channel k - XPW
  auto prefetch_piece = [&](int k) {
    if (k < XPW) {
      const __amdgpu_buffer_rsrc_t rx = plane_rsrc(xp[k], xsb[k], pf_n, xnb[k] & pf_live);
#pragma unroll
      for (int j = 0; j < XSLOTS; ++j)
        xr[k][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, xoff[j], 0, 0));
    } else if (k < XPW + GPW) {
      const int c = k - XPW;
      const __amdgpu_buffer_rsrc_t rg = plane_rsrc(gp[c], gsb, pf_n, gnb[c] & pf_live);
#pragma unroll
      for (int j = 0; j < GSLOTS; ++j)
        gr[c][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rg, goff[j], 0, 0));
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
#pragma unroll
      for (int j = 0; j < GSLOTS; ++j) {
        if (64 * (j + 1) <= GE || lane + 64 * j < GE) g_s[(GPW * wave + c) * GSTR + lane + 64 * j] = gr[c][j];
        gsum[c] += gr[c][j];   // out-of-range lanes loaded zeros
      }
  };

  f32x16 acc[TAPS];
#pragma unroll
  for (int t = 0; t < TAPS; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  const float* const a_base = g_s + (mt * 32 + l32) * GSTR + half + 2 * grp;
  const float* const b_base = x_s + (ct * 32 + l32) * XSTR + (half + 2 * grp) * SL;
  static_assert(TW % (2 * KG) == 0 && GE % (2 * KG) == 0, "pixel-pair interleave must tile a row");
  constexpr int NPAIR = GE / (2 * KG);                     // pixel pairs per wave and tile
  constexpr int NPIECE = XPW + GPW;
  constexpr int PPS = (NPIECE + NPAIR - 1) / NPAIR;        // load pieces issued after each pair

  int t = blockIdx.x;
  prefetch_setup(t);
#pragma unroll
  for (int k = 0; k < NPIECE; ++k) prefetch_piece(k);
  commit();
  __syncthreads();
  for (; t < a.n_tiles; t += nsplit) {
    prefetch_setup(t + nsplit);
#pragma unroll
    for (int jj = 0; jj < NPAIR; ++jj) {   // this wave's pixel pairs: pixels 2 KG jj + 2 grp, + 1
      const int row = (2 * KG * jj) / TW, px = (2 * KG * jj) % TW;
      const float av = a_base[row * TW + px];
#pragma unroll
      for (int ky = 0; ky < KH; ++ky)
#pragma unroll
        for (int kx = 0; kx < KS; ++kx) {
          const float bv = b_base[(row * SL + ky) * IW + px * SL + kx];
          acc[ky * KS + kx] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[ky * KS + kx], 0, 0, 0);
        }
#pragma unroll
      for (int q = 0; q < PPS; ++q) prefetch_piece(jj * PPS + q);
      WG_SCHED_FENCE();
    }
    __syncthreads();                     // every wave is done reading this tile
    commit();                            // (a tile past the end committed zeros: nobody reads them)
    __syncthreads();
  }

  // Bias gradient: one workgroup per output-channel block (input-channel block 0) writes the
  // sums of its split; lanes are added in a fixed butterfly order.
  if (a.bias_partial != nullptr && cib == 0) {
#pragma unroll
    for (int c = 0; c < GPW; ++c) {
      float v = gsum[c];
#pragma unroll
      for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
      if (lane == 0) a.bias_partial[(long long)blockIdx.x * a.cout_pad + co0 + GPW * wave + c] = v;
    }
  }

  // Sum the KG waves that share a corner: per tap, groups 1 .. KG-1 park their tile in LDS and
  // group 0 adds them in ascending order, then stores the tap of the partial block
  // [split][tap][co][ci]; register r of a tile is row (r&3) + 8 (r>>2) + 4 half, column l32.
  float* const red = lds;   // [KG-1][MT*CT][16][64]; the staging buffers are free after the last barrier
  float* const pbase = a.partial + (long long)blockIdx.x * TAPS * a.cout_pad * a.cin_pad;
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
      float* const pt = pbase + (long long)tp * a.cout_pad * a.cin_pad + ci0 + ct * 32 + l32;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = co0 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        pt[(long long)co * a.cin_pad] = v[r];
      }
    }
    __syncthreads();
  }
}

// 7x7 (stride 1): 49 taps of a 32 x 32 (Cout x Cin) block are 49 accumulator tiles -- too many for one
// wave, and re-staging the pixel tile once per tap row makes a kernel wait on its loads.  Here wave w of
// a workgroup owns tap row ky = w (7 accumulator tiles, kx = 0..6); all waves read the same staged tile:
// TH x 32 output pixels of 32 gradient channels and the (TH + 6) x 38 halo tile of 32 input channels, so
// one staged tile feeds 7 x 7 x TH x 16 MFMAs.  The eighth wave only helps staging.  No cross-wave
// reduction is needed; splits over the pixel axis and the second kernel are as above.
struct W7 {
  static constexpr int TH = tile_rows(7, 1), IH = TH + 6, IW = 38, XE = IH * IW, XSTR = XE | 1, XSLOTS = (XE + 63) / 64;
  static constexpr int GE = TH * 32, GSTR = GE + 1, GSLOTS = GE / 64;
  static constexpr int CPW = 4;   // channels of x and of g staged per wave (8 waves x 4 = 32)
  static constexpr int LDS_BYTES = (32 * XSTR + 32 * GSTR) * 4;
};

// NARROW (Cin <= 8: the first layer of a motion-estimation level, 8 -> 32): the 32 columns of a matrix-core block are
// 8 input channels x 4 horizontally adjacent taps instead of 32 channels at one tap -- column 8 j + ci reads the
// staged tile j pixels to the right -- so a tap row takes 2 accumulator tiles (kx = 0..3, 4..6 and one idle
// column group) instead of 7 with three quarters of every block multiplying zeros.
template <bool NARROW>
__global__ __launch_bounds__(512) void conv_wgrad7_kernel(const WgradArgs a) {
  constexpr int TH = W7::TH, IW = W7::IW, XE = W7::XE, XSTR = W7::XSTR, XSLOTS = W7::XSLOTS;
  constexpr int GSTR = W7::GSTR, GSLOTS = W7::GSLOTS, CPW = W7::CPW;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* const x_s = lds;                 // [32][XSTR]
  float* const g_s = lds + 32 * XSTR;     // [32][GSTR]

  const int tid = threadIdx.x;
  const int lane = tid & 63, half = lane >> 5, l32 = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cob = blockIdx.y / a.n_cib, cib = blockIdx.y - cob * a.n_cib;
  const int co0 = cob * 32, ci0 = cib * 32;
  const int nsplit = gridDim.x;

  const int cout = a.cout;
  const unsigned xplane_bytes = 4u * (unsigned)(a.h * a.w), gplane_bytes = 4u * (unsigned)(a.ho * a.wo);
  const float in_slope = a.in_slope;
  const int per_n = a.tiles_x * a.tiles_y;

  unsigned long long xp[CPW], xsb[CPW], gp[CPW];
  unsigned xnb[CPW], gnb[CPW];
#pragma unroll
  for (int c = 0; c < CPW; ++c) source_plane(a, ci0 + CPW * wave + c, xplane_bytes, &xp[c], &xsb[c], &xnb[c]);
  const unsigned long long gsb = (unsigned long long)cout * (unsigned long long)gplane_bytes;
#pragma unroll
  for (int c = 0; c < CPW; ++c) {
    const int co = co0 + CPW * wave + c;
    const bool haveg = co < cout;
    gp[c] = uniform64(reinterpret_cast<unsigned long long>(a.g) + (haveg ? (unsigned long long)co * gplane_bytes : 0ull));
    gnb[c] = __builtin_amdgcn_readfirstlane(haveg ? gplane_bytes : 0u);
  }

  float xr[CPW][XSLOTS], gr[CPW][GSLOTS], gsum[CPW];
#pragma unroll
  for (int c = 0; c < CPW; ++c) gsum[c] = 0.f;

  unsigned xoff[XSLOTS], goff[GSLOTS];
  int pf_n = 0;
  unsigned pf_live = 0;
  auto prefetch_setup = [&](int t) {
    pf_live = t < a.n_tiles ? 0xffffffffu : 0u;
    const int n = t / per_n, r = t - n * per_n;
    const int ty = r / a.tiles_x, tx = r - ty * a.tiles_x;
    const int oy0 = ty * TH, ox0 = tx * 32;
    pf_n = n;
#pragma unroll
    for (int j = 0; j < XSLOTS; ++j) {
      const int e = lane + 64 * j;
      const int iy = e / IW, ix = e - iy * IW;
      const int gy = oy0 - 3 + iy, gx = ox0 - 3 + ix;
      const bool ok = e < XE && gy >= 0 && gy < a.h && gx >= 0 && gx < a.w;
      xoff[j] = ok ? 4u * (unsigned)(gy * a.w + gx) : 0x80000000u;
    }
#pragma unroll
    for (int j = 0; j < GSLOTS; ++j) {
      const int oy = oy0 + 2 * j + (lane >> 5), ox = ox0 + (lane & 31);
      goff[j] = (oy < a.ho && ox < a.wo) ? 4u * (unsigned)(oy * a.wo + ox) : 0x80000000u;
    }
  };
  auto prefetch_piece = [&](int k) {   // k < CPW: x channel k; CPW <= k < 2 CPW: g channel k - CPW
    if (k < CPW) {
      const __amdgpu_buffer_rsrc_t rx = plane_rsrc(xp[k], xsb[k], pf_n, xnb[k] & pf_live);
#pragma unroll
      for (int j = 0; j < XSLOTS; ++j)
        xr[k][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, xoff[j], 0, 0));
    } else if (k < 2 * CPW) {
      const int c = k - CPW;
      const __amdgpu_buffer_rsrc_t rg = plane_rsrc(gp[c], gsb, pf_n, gnb[c] & pf_live);
#pragma unroll
      for (int j = 0; j < GSLOTS; ++j)
        gr[c][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rg, goff[j], 0, 0));
    }
  };

  auto commit = [&]() {
#pragma unroll
    for (int c = 0; c < CPW; ++c) {
      float* xd = x_s + (CPW * wave + c) * XSTR + lane;
#pragma unroll
      for (int j = 0; j < XSLOTS; ++j)
        if (64 * (j + 1) <= XE || lane + 64 * j < XE) xd[64 * j] = fmaxf(xr[c][j], xr[c][j] * in_slope);
      float* gd = g_s + (CPW * wave + c) * GSTR + lane;
#pragma unroll
      for (int j = 0; j < GSLOTS; ++j) {
        gd[64 * j] = gr[c][j];
        gsum[c] += gr[c][j];   // out-of-range lanes loaded zeros
      }
    }
  };

  constexpr int NACC = NARROW ? 2 : 7, KSTEP = NARROW ? 4 : 1;
  f32x16 acc[NACC];
#pragma unroll
  for (int t = 0; t < NACC; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  const float* const a_base = g_s + l32 * GSTR + half;
  // tap row ky = wave; NARROW: column l32 = input channel l32 & 7 at horizontal tap offset l32 >> 3
  const float* const b_base = NARROW ? x_s + (l32 & 7) * XSTR + half + wave * IW + (l32 >> 3)
                                     : x_s + l32 * XSTR + half + wave * IW;

  int t = blockIdx.x;
  prefetch_setup(t);
#pragma unroll
  for (int k = 0; k < 2 * CPW; ++k) prefetch_piece(k);
  commit();
  __syncthreads();
  for (; t < a.n_tiles; t += nsplit) {
    prefetch_setup(t + nsplit);
    if (wave < 7) {
#pragma unroll
      for (int row = 0; row < TH; ++row) {
#pragma unroll
        for (int px = 0; px < 32; px += 2) {
          const float av = a_base[row * 32 + px];
#pragma unroll
          for (int kx = 0; kx < NACC; ++kx)
            acc[kx] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b_base[row * IW + px + kx * KSTEP], acc[kx], 0, 0, 0);
          if (px % 16 == 0) prefetch_piece(row * 2 + px / 16);   // 2 pieces per row: 8 = 2 CPW in all
          WG_SCHED_FENCE();
        }
      }
    } else {
#pragma unroll
      for (int k = 0; k < 2 * CPW; ++k) prefetch_piece(k);
    }
    __syncthreads();                     // every wave is done reading this tile
    commit();                            // (a tile past the end committed zeros: nobody reads them)
    __syncthreads();
  }

  if (a.bias_partial != nullptr && cib == 0) {
#pragma unroll
    for (int c = 0; c < CPW; ++c) {
      float v = gsum[c];
#pragma unroll
      for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
      if (lane == 0) a.bias_partial[(long long)blockIdx.x * a.cout_pad + co0 + CPW * wave + c] = v;
    }
  }
  if (wave < 7) {
    float* const pbase = a.partial + (long long)blockIdx.x * 49 * a.cout_pad * a.cin_pad;
    if constexpr (NARROW) {
      // (channels 8..31 of the padded block are never read back: the reduction stops at Cin)
#pragma unroll
      for (int grp = 0; grp < 2; ++grp) {
        const int kx = 4 * grp + (l32 >> 3);
        if (kx < 7) {
          float* const pt = pbase + (long long)(wave * 7 + kx) * a.cout_pad * a.cin_pad + ci0 + (l32 & 7);
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int co = co0 + (r & 3) + 8 * (r >> 2) + 4 * half;
            pt[(long long)co * a.cin_pad] = acc[grp][r];
          }
        }
      }
    } else {
#pragma unroll
      for (int kx = 0; kx < 7; ++kx) {
        float* const pt = pbase + (long long)(wave * 7 + kx) * a.cout_pad * a.cin_pad + ci0 + l32;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int co = co0 + (r & 3) + 8 * (r >> 2) + 4 * half;
          pt[(long long)co * a.cin_pad] = acc[kx][r];
        }
      }
    }
  }
}


// ------------------------------------------------------------------ layers with <= 4 output channels
// The last layer of every motion-estimation level (16 -> 2, 7x7) fills 2 of the 32 rows of a matrix-core block: the
// 7x7 kernel above took 1.67 ms for it at 32 x 256 x 256 (3.9 TFLOP/s) where the layer's bytes need 0.05 ms; this
// form takes 0.46-0.49 ms (bound by the round trips of staging a tile, two workgroups per CU).  Here the products run on the vector ALUs: thread (ci, ky)
// keeps gW[co][ci][ky][0..KS) for every co in registers and walks the rows of a 4 x 32 pixel tile with a sliding
// window of KS input values -- one LDS read of x, one 16-byte broadcast read of the gradients and KS x cout FMAs per
// pixel.  2 or 4 groups of threads (Cin x KS <= 128 or <= 64) share the rows of a tile and count
// as separate slabs of the pixel axis.  Workgroup `s` walks tiles s, s + S, ...; the partial sums go through the
// same fixed-order reduction kernel as the other forms (layout [slab][tap][co][ci], no padding).
template <int KS>
struct WThin {
  static constexpr int TH = 4, TW = 32, PAD = KS / 2, ROWS = TH + KS - 1, COLS = TW + KS - 1;
  // LDS strides: rows KS banks apart and channels one bank apart, so that the threads (ci, ky) of a wave read
  // different banks
  static constexpr int RS = COLS + ((KS - COLS % 32) + 32) % 32;
  static constexpr int CS = ROWS * RS + ((1 - (ROWS * RS) % 32) + 32) % 32;
  static constexpr int MAX_CIN = 64;
  static constexpr int lds_bytes(int cin) { return cin * CS * 4; }
};

template <int KS>
__global__ __launch_bounds__(256) void conv_wgrad_thin_kernel(const WgradArgs a, int parts) {
  using T = WThin<KS>;
  extern __shared__ float xs[];                      // [cin][ROWS][COLS] with strides CS, RS
  __shared__ __attribute__((aligned(16))) float gs[T::TH * T::TW * 4];   // [pixel][co]
  __shared__ const float* plane[T::MAX_CIN];         // channel -> plane of sample 0
  __shared__ long long plane_bs[T::MAX_CIN];
  const int tid = threadIdx.x;
  const int slots = 256 / parts;
  const int role = tid % slots, part = tid / slots;
  const int nroles = a.cin * KS;
  const bool active = role < nroles;
  const int ci = active ? role / KS : 0, ky = active ? role - ci * KS : 0;
  if (tid < a.cin) {
    int c = tid, s = 0;
    while (s + 1 < a.n_src && c >= a.src[s].channels) { c -= a.src[s].channels; ++s; }
    plane[tid] = a.src[s].ptr + (long long)c * a.src[s].chan_stride;
    plane_bs[tid] = a.src[s].batch_stride;
  }
  float acc[KS][4];
  float gsum[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int kx = 0; kx < KS; ++kx)
#pragma unroll
    for (int co = 0; co < 4; ++co) acc[kx][co] = 0.f;
  const int tiles_per_sample = a.tiles_x * a.tiles_y;
  const long long gplane = (long long)a.ho * a.wo;
  for (int t = blockIdx.x; t < a.n_tiles; t += gridDim.x) {
    const int n = t / tiles_per_sample, rem = t - n * tiles_per_sample;
    const int ty = rem / a.tiles_x, tx = rem - ty * a.tiles_x;
    const int oy0 = ty * T::TH, ox0 = tx * T::TW;
    __syncthreads();   // the previous tile has been consumed (first tile: the plane table is written)
    float gv[2];   // the gradient tile: two values per thread, requested before the input tile and stored after it
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int idx = tid + 256 * u;
      const int co = idx / (T::TH * T::TW), px = idx - co * (T::TH * T::TW);
      const int oy = oy0 + px / T::TW, ox = ox0 + px % T::TW;
      gv[u] = (co < a.cout && oy < a.ho && ox < a.wo) ? a.g[((long long)n * a.cout + co) * gplane + (long long)oy * a.wo + ox] : 0.f;
    }
    // sixteen requests per thread in flight before the first is used (one at a time, the tile's ~50 dependent
    // round trips to memory were 90 % of the kernel)
    constexpr int U = 16;
    const int total = a.cin * T::ROWS * T::COLS;
    for (int base = tid; base < total; base += 256 * U) {
      float v[U];
      int la[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int idx = base + 256 * u;
        const int c = idx / (T::ROWS * T::COLS), r2 = idx - c * (T::ROWS * T::COLS);
        const int ry = r2 / T::COLS, rx = r2 - ry * T::COLS;
        const int iy = oy0 - T::PAD + ry, ix = ox0 - T::PAD + rx;
        const bool inside = idx < total && iy >= 0 && iy < a.h && ix >= 0 && ix < a.w;
        la[u] = idx < total ? c * T::CS + ry * T::RS + rx : -1;
        v[u] = inside ? plane[c][n * plane_bs[c] + (long long)iy * a.w + ix] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (la[u] >= 0) xs[la[u]] = dvc::leaky(v[u], a.in_slope);
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int idx = tid + 256 * u;
      const int co = idx / (T::TH * T::TW), px = idx - co * (T::TH * T::TW);
      gs[px * 4 + co] = gv[u];
    }
    __syncthreads();
    if (active) {
      for (int r = part; r < T::TH; r += parts) {
        const float* const xr = xs + ci * T::CS + (r + ky) * T::RS;
        const float4* const gr = reinterpret_cast<const float4*>(gs) + r * T::TW;
        float win[KS];
#pragma unroll
        for (int k = 0; k < KS - 1; ++k) win[k] = xr[k];
#pragma unroll
        for (int ox = 0; ox < T::TW; ++ox) {
          win[(ox + KS - 1) % KS] = xr[ox + KS - 1];
          const float4 g4 = gr[ox];
#pragma unroll
          for (int kx = 0; kx < KS; ++kx) {
            const float xv = win[(ox + kx) % KS];
            acc[kx][0] += g4.x * xv; acc[kx][1] += g4.y * xv; acc[kx][2] += g4.z * xv; acc[kx][3] += g4.w * xv;
          }
          if (role == 0) { gsum[0] += g4.x; gsum[1] += g4.y; gsum[2] += g4.z; gsum[3] += g4.w; }
        }
      }
    }
  }
  const long long slab = (long long)blockIdx.x * parts + part;
  if (active) {
    float* const pb = a.partial + slab * (KS * KS) * a.cout_pad * a.cin_pad;
#pragma unroll
    for (int kx = 0; kx < KS; ++kx)
#pragma unroll
      for (int co = 0; co < 4; ++co)
        if (co < a.cout) pb[((long long)(ky * KS + kx) * a.cout_pad + co) * a.cin_pad + ci] = acc[kx][co];
    if (role == 0 && a.bias_partial != nullptr) {
#pragma unroll
      for (int co = 0; co < 4; ++co)
        if (co < a.cout) a.bias_partial[slab * a.cout_pad + co] = gsum[co];
    }
  }
}


// gW[co][ci][tap] = sum over splits of partial[split][tap][co][ci]; gB[co] = sum over splits of
// bias_partial[split][co] (indices past the weight's).  A block handles 64 results; its four waves each
// add every fourth split (independent loads in flight instead of one long dependent chain) and the four
// sums are combined through LDS in a fixed order, so the result does not depend on scheduling.
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* partial, float* gw, int nsplit, int taps,
                                                           int cout, int cin, int cout_pad, int cin_pad,
                                                           const float* bias_partial, float* gb) {
  __shared__ float part[4][64];
  const long long total = (long long)cout * taps * cin;
  const long long sstride = (long long)taps * cout_pad * cin_pad;
  const long long total_all = total + (gb != nullptr ? cout : 0);
  const int lane = threadIdx.x & 63, kg = threadIdx.x >> 6;
  for (long long i0 = blockIdx.x * 64LL; i0 < total_all; i0 += (long long)gridDim.x * 64) {   // block-uniform
    const long long i = i0 + lane;
    const float* p = nullptr;
    long long step = 0;
    long long dst = -1;
    if (i < total) {
      const int ci = (int)(i % cin);
      const long long q = i / cin;
      const int tap = (int)(q % taps), co = (int)(q / taps);
      p = partial + ((long long)tap * cout_pad + co) * cin_pad + ci;
      step = sstride;
      dst = ((long long)co * cin + ci) * taps + tap;
    } else if (i < total_all) {
      p = bias_partial + (i - total);
      step = cout_pad;
    }
    float s = 0.f;
    if (p != nullptr)
      for (int k = kg; k < nsplit; k += 4) s += p[k * step];
    part[kg][lane] = s;
    __syncthreads();
    if (kg == 0 && p != nullptr) {
      const float v = ((part[0][lane] + part[1][lane]) + part[2][lane]) + part[3][lane];
      if (dst >= 0) gw[dst] = v;
      else gb[i - total] = v;
    }
    __syncthreads();
  }
}

struct Plan {
  int ho, wo, th, tiles_x, tiles_y, n_tiles, cout_pad, cin_pad, n_cib, blocks, nsplit, taps, mt, ct;
  int tw;           // pixels per tile row (32; 16 for 3x3 stride-1 layers on planes at most 16 wide)
  int thin_parts;   // > 0: the vector-ALU kernel for <= 4 output channels, with this many thread groups per tile
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
  p->tw = (ks == 3 && stride == 1 && p->wo <= 16 && cout > 4) ? 16 : 32;
  p->th = tile_rows(ks, stride) * (32 / p->tw);
  p->tiles_x = (p->wo + p->tw - 1) / p->tw;
  p->tiles_y = (p->ho + p->th - 1) / p->th;
  const long long nt = (long long)batch * p->tiles_x * p->tiles_y;
  DVC_REQUIRE(nt < (1LL << 30), "%s: too many tiles", who);
  p->n_tiles = (int)nt;
  p->thin_parts = 0;
  // (Cin x KS <= 128: at least two thread groups per tile.  With 64 input channels -- the 64 -> 3 reconstruction layer --
  // the matrix-core form is as fast, 650 against 680 us at 32 x 256 x 256, although it pads 3 channels to 32)
  if (cout <= 4 && stride == 1 && ks != 1 && cin * ks <= 128) {
    const int roles = cin * ks;
    p->thin_parts = roles <= 64 ? 4 : 2;
    p->mt = p->ct = 1;
    p->cout_pad = cout;
    p->cin_pad = cin;
    p->n_cib = 1;
    p->blocks = 1;
    p->taps = ks * ks;
    // two workgroups per CU (more of them hide the staging round trips better, but every workgroup is a slab of
    // partial sums for the reduction kernel, which has only Cout x Cin x taps / 64 workgroups to add them up with:
    // 5 per CU measured 535 us against 455 us for 16 -> 2 7x7 at 32 x 256 x 256)
    const int wgs = p->n_tiles < 512 ? p->n_tiles : 512;
    p->nsplit = wgs * p->thin_parts;
    return DVC_OK;
  }
  p->mt = (cout > 32 && ks != 7) ? 2 : 1;   // 7x7: 32 x 32 blocks, one tap row per wave
  p->ct = (cin > 32 && ks != 7) ? 2 : 1;
  const int cob = 32 * p->mt, cib = 32 * p->ct;
  p->cout_pad = (cout + cob - 1) / cob * cob;
  p->cin_pad = (cin + cib - 1) / cib * cib;
  p->n_cib = p->cin_pad / cib;
  p->blocks = (p->cout_pad / cob) * p->n_cib;
  p->taps = ks * ks;
  const int wg_per_split = p->blocks;
  int s = (256 + wg_per_split - 1) / wg_per_split;   // one 8-wave workgroup fits a CU: about one round of them
  if (s > p->n_tiles) s = p->n_tiles;
  if (s < 1) s = 1;
  p->nsplit = s;
  return DVC_OK;
}

template <int KS, int S, int MT, int CT, int TW = 32>
int launch(const WgradArgs& a, const Plan& p, hipStream_t st) {
  using G = WGeo<KS, S, MT, CT, TW>;
  // LDS above the 64 KiB default needs an opt-in per kernel and per device; calls may come from
  // several host threads, so the done-mask is atomic (setting the attribute twice is harmless)
  static std::atomic<unsigned long long> attr_done{0};
  int device = 0;
  if (hipGetDevice(&device) != hipSuccess || device < 0 || device > 63) device = 0;
  const unsigned long long device_bit = 1ull << device;
  if (!(attr_done.load(std::memory_order_acquire) & device_bit)) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_kernel<KS, S, MT, CT, TW>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES);
    if (e != hipSuccess) {
      dvc::set_error("hipFuncSetAttribute(wgrad %d,%d): %s", KS, S, hipGetErrorString(e));
      return DVC_EHIP;
    }
    attr_done.fetch_or(device_bit, std::memory_order_release);
  }
  dim3 grid(p.nsplit, p.blocks, 1);
  hipLaunchKernelGGL((conv_wgrad_kernel<KS, S, MT, CT, TW>), grid, dim3(512), G::LDS_BYTES, st, a);
  return dvc::check_launch("conv_wgrad_kernel");
}

template <bool NARROW>
int launch7(const WgradArgs& a, const Plan& p, hipStream_t st) {
  static std::atomic<unsigned long long> attr_done{0};
  int device = 0;
  if (hipGetDevice(&device) != hipSuccess || device < 0 || device > 63) device = 0;
  const unsigned long long device_bit = 1ull << device;
  if (!(attr_done.load(std::memory_order_acquire) & device_bit)) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad7_kernel<NARROW>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, W7::LDS_BYTES);
    if (e != hipSuccess) {
      dvc::set_error("hipFuncSetAttribute(wgrad 7x7): %s", hipGetErrorString(e));
      return DVC_EHIP;
    }
    attr_done.fetch_or(device_bit, std::memory_order_release);
  }
  hipLaunchKernelGGL(conv_wgrad7_kernel<NARROW>, dim3(p.nsplit, p.blocks, 1), dim3(512), W7::LDS_BYTES, st, a);
  return dvc::check_launch("conv_wgrad7_kernel");
}

template <int KS>
int launch_thin(const WgradArgs& a, const Plan& p, hipStream_t st) {
  // (make_plan admits Cin * KS <= 128)
  constexpr int kMaxCin = 128 / KS;
  static_assert(WThin<KS>::lds_bytes(kMaxCin) + 4 * 4 * 32 * 4 + 64 * 16 <= 64 * 1024, "inside the default LDS limit");
  hipLaunchKernelGGL(conv_wgrad_thin_kernel<KS>, dim3(p.nsplit / p.thin_parts), dim3(256), WThin<KS>::lds_bytes(a.cin), st, a,
                     p.thin_parts);
  return dvc::check_launch("conv_wgrad_thin_kernel");
}

template <int KS, int S>
int launch_block(const WgradArgs& a, const Plan& p, hipStream_t st) {
  if constexpr (KS == 3 && S == 1) {
    if (p.tw == 16) {
      if (p.mt == 2) return p.ct == 2 ? launch<3, 1, 2, 2, 16>(a, p, st) : launch<3, 1, 2, 1, 16>(a, p, st);
      return p.ct == 2 ? launch<3, 1, 1, 2, 16>(a, p, st) : launch<3, 1, 1, 1, 16>(a, p, st);
    }
  }
  if (p.mt == 2) return p.ct == 2 ? launch<KS, S, 2, 2>(a, p, st) : launch<KS, S, 2, 1>(a, p, st);
  return p.ct == 2 ? launch<KS, S, 1, 2>(a, p, st) : launch<KS, S, 1, 1>(a, p, st);
}

}  // namespace

extern "C" int64_t dvc_conv2d_wgrad_scratch_floats(int32_t cout, int32_t cin, int32_t ksize, int32_t stride,
                                                   int32_t batch, int32_t in_h, int32_t in_w) {
  Plan p;
  if (make_plan(cout, cin, ksize, stride, batch, in_h, in_w, &p, "dvc_conv2d_wgrad_scratch_floats")) return DVC_EINVAL;
  return (int64_t)p.nsplit * p.taps * p.cout_pad * p.cin_pad + (int64_t)p.nsplit * p.cout_pad;
}

extern "C" int dvc_conv2d_wgrad(const DvcSrc* srcs, int32_t n_src, int32_t batch, int32_t in_h, int32_t in_w,
                                int32_t in_act, float in_slope, const float* grad_out, int32_t out_channels,
                                int32_t ksize, int32_t stride, float* scratch, float* grad_weight, float* grad_bias,
                                void* stream) {
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
  a.bias_partial = grad_bias != nullptr ? scratch + (long long)p.nsplit * p.taps * p.cout_pad * p.cin_pad : nullptr;
  a.cout_pad = p.cout_pad;
  a.cin_pad = p.cin_pad;
  a.n_cib = p.n_cib;
  a.tiles_x = p.tiles_x;
  a.tiles_y = p.tiles_y;
  a.n_tiles = p.n_tiles;
  a.in_slope = in_act ? in_slope : 1.0f;
  hipStream_t st = static_cast<hipStream_t>(stream);
  int rc;
  if (p.thin_parts > 0) rc = ksize == 3 ? launch_thin<3>(a, p, st) : launch_thin<7>(a, p, st);
  else if (ksize == 3) rc = stride == 1 ? launch_block<3, 1>(a, p, st) : launch_block<3, 2>(a, p, st);
  else if (ksize == 1) rc = stride == 1 ? launch_block<1, 1>(a, p, st) : launch_block<1, 2>(a, p, st);
  else rc = cin <= 8 ? launch7<true>(a, p, st) : launch7<false>(a, p, st);
  if (rc) return rc;
  const long long total = (long long)out_channels * cin * p.taps + out_channels;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(dvc::stream_grid(total, 64)), dim3(256), 0, st, scratch, grad_weight,
                     p.nsplit, p.taps, out_channels, cin, p.cout_pad, p.cin_pad, a.bias_partial, grad_bias);
  return dvc::check_launch("wgrad_reduce_kernel");
}
