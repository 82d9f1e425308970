// Direct convolution for layers with at most 4 output channels (the last layer of each SpyNet level,
// 16 -> 2 at 7x7, and the 64 -> 3 reconstruction layer): on the matrix cores such a layer keeps 2 or 3
// of 32 MFMA rows busy, here every multiply is useful.
//
// A workgroup of 8 x ROWS lanes computes a 32-pixel-wide, ROWS-high tile: a lane owns four horizontally adjacent
// pixels and all output channels of them.  Input channels go through LDS a few at a time as (ROWS + k - 1) x
// (32 + k - 1) halo tiles with a 16-byte-aligned row pitch (two buffers: the next chunk is loaded while the current
// one is multiplied).  Per tap row a lane reads its 4 + k - 1 inputs with 16-byte LDS reads and does 4 k Cout FMAs
// on them -- the kernel is bound by the vector ALUs, not by LDS reads (one pixel per lane: one read per Cout FMAs).
// The weights of a channel are wave-uniform and are read as scalars straight from the packed slabs the MFMA kernel
// uses ([chunk][(ci - c0) * taps + tap][32 output channels], zero filled; staging them in LDS for broadcast reads was
// measured: faster for the one-wave tiles, slower for the large ones, no gain in sum).  Every output value is the same fmaf
// chain (channel major, tap minor) as on the matrix cores.  Same fused prologue / epilogue as the MFMA kernel for
// the cases these layers use: input and output LeakyReLU, bias, up to two residuals.
#include "conv_args.h"

namespace dvc {
namespace {

constexpr int T_TW = 32, T_PX = 4;   // tile width, pixels per lane
constexpr int thin_chunk(int ks) { return ks == 7 ? 2 : 4; }   // input channels per LDS buffer

template <int KS, int CO, int ROWS>
__global__ __launch_bounds__(8 * ROWS) void conv_thin_kernel(const ConvArgs a, int ck_packed, int slab) {
  constexpr int NTHR = 8 * ROWS;
  constexpr int T_CK = thin_chunk(KS);
  constexpr int PAD = KS / 2, IH = ROWS + KS - 1, IWV = T_TW + KS - 1, IW = (IWV + 3) / 4 * 4, IE = IH * IW, TAPS = KS * KS;
  constexpr int SPC = (IE + NTHR - 1) / NTHR;   // staging slots per thread and channel
  constexpr int NV = T_PX + KS - 1;              // inputs a lane needs per tap row
  __shared__ __attribute__((aligned(16))) float in_s[2][T_CK * IE];
  const int tid = threadIdx.x, tx = tid & 7, ty = tid >> 3;
  const int n = blockIdx.z;
  const int oy0 = blockIdx.y * ROWS, ox0 = blockIdx.x * T_TW;
  const float* const sp = a.src[0].ptr + (long long)n * a.src[0].batch_stride;
  const long long cs = a.src[0].chan_stride;
  const int cin = a.cin;
  const float in_slope = a.in_slope;

  // halo position tid + NTHR * j of a channel: its offset inside the plane (-1: padding, or the pitch's filler)
  float stage[T_CK][SPC];
  int soff[SPC];
#pragma unroll
  for (int j = 0; j < SPC; ++j) {
    const int r = tid + NTHR * j;
    const int iy = r / IW, ix = r - iy * IW;
    const int gy = oy0 - PAD + iy, gx = ox0 - PAD + ix;
    const bool ok = r < IE && ix < IWV && gy >= 0 && gy < a.h && gx >= 0 && gx < a.w;
    soff[j] = ok ? gy * a.w + gx : -1;
  }
  auto load_chunk = [&](int c0) __attribute__((always_inline)) {
#pragma unroll
    for (int c = 0; c < T_CK; ++c) {
      const bool have = c0 + c < cin;                        // uniform
      const float* const plane = sp + (long long)(have ? c0 + c : 0) * cs;
#pragma unroll
      for (int j = 0; j < SPC; ++j) stage[c][j] = (have && soff[j] >= 0) ? plane[soff[j]] : 0.f;
    }
  };
  auto store_chunk = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
    for (int c = 0; c < T_CK; ++c)
#pragma unroll
      for (int j = 0; j < SPC; ++j)
        if (NTHR * (j + 1) <= IE || tid + NTHR * j < IE) in_s[buf][c * IE + tid + NTHR * j] = leaky(stage[c][j], in_slope);   // slope 1 = identity
  };

  float acc[CO][T_PX];
#pragma unroll
  for (int co = 0; co < CO; ++co)
#pragma unroll
    for (int p = 0; p < T_PX; ++p) acc[co][p] = 0.f;

  const int nchunks = (cin + T_CK - 1) / T_CK;
  load_chunk(0);
  store_chunk(0);
  __syncthreads();
  for (int k = 0; k < nchunks; ++k) {
    const int cur = k & 1;
    if (k + 1 < nchunks) load_chunk((k + 1) * T_CK);
    const int c_end = min(T_CK, cin - k * T_CK);   // uniform
    for (int c = 0; c < c_end; ++c) {
      const int ci = k * T_CK + c;
      // weights of channel ci, all taps: rows of 32 output channels in the packed slab (uniform address)
      const float* wrow = a.wpacked + (long long)(ci / ck_packed) * slab + (long long)((ci % ck_packed) * TAPS) * 32;
      const float* xin = in_s[cur] + c * IE + ty * IW + T_PX * tx;
#pragma unroll
      for (int ky = 0; ky < KS; ++ky) {
        float v[(NV + 3) / 4 * 4];
#pragma unroll
        for (int q = 0; q < (NV + 3) / 4; ++q) {   // (the pitch covers whole 16-byte pieces)
          const float4 t = *reinterpret_cast<const float4*>(xin + ky * IW + 4 * q);
          v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w;
        }
#pragma unroll
        for (int kx = 0; kx < KS; ++kx)
#pragma unroll
          for (int co = 0; co < CO; ++co) {
            const float wv = wrow[(ky * KS + kx) * 32 + co];
#pragma unroll
            for (int p = 0; p < T_PX; ++p) acc[co][p] = fmaf(wv, v[kx + p], acc[co][p]);
          }
      }
    }
    if (k + 1 < nchunks) store_chunk(cur ^ 1);   // that buffer was last read before the previous barrier
    __syncthreads();
  }

  const int oy = oy0 + ty, ox = ox0 + T_PX * tx;
  if (oy >= a.ho || ox >= a.wo) return;
  const long long pix = (long long)oy * a.wo + ox;
  const bool whole = a.vec4 && ox + T_PX <= a.wo;   // 16-byte rows: the lane's four pixels as one access
#pragma unroll
  for (int co = 0; co < CO; ++co) {
    if (co >= a.cout) break;
    const float bias = a.bias != nullptr ? a.bias[co] : 0.f;
    float v[T_PX];
#pragma unroll
    for (int p = 0; p < T_PX; ++p) v[p] = leaky(acc[co][p] + bias, a.out_slope);
    float* const op = a.out + (long long)n * a.obs + co * a.ocs + pix;
    const float* const rp = a.res != nullptr ? a.res + (long long)n * a.rbs + co * a.rcs + pix : nullptr;
    const float* const r2p = a.res2 != nullptr ? a.res2 + (long long)n * a.r2bs + co * a.r2cs + pix : nullptr;
    if (whole) {
      if (rp != nullptr) {
        const float4 r = *reinterpret_cast<const float4*>(rp);
        v[0] = __fadd_rn(v[0], r.x); v[1] = __fadd_rn(v[1], r.y); v[2] = __fadd_rn(v[2], r.z); v[3] = __fadd_rn(v[3], r.w);
        if (r2p != nullptr) {
          const float4 r2 = *reinterpret_cast<const float4*>(r2p);
          v[0] = __fadd_rn(v[0], r2.x); v[1] = __fadd_rn(v[1], r2.y); v[2] = __fadd_rn(v[2], r2.z); v[3] = __fadd_rn(v[3], r2.w);
        }
      }
      *reinterpret_cast<float4*>(op) = make_float4(v[0], v[1], v[2], v[3]);
    } else {
#pragma unroll
      for (int p = 0; p < T_PX; ++p) {
        if (ox + p >= a.wo) break;
        float t = v[p];
        if (rp != nullptr) {
          t = __fadd_rn(t, rp[p]);
          if (r2p != nullptr) t = __fadd_rn(t, r2p[p]);
        }
        op[p] = t;
      }
    }
  }
}

// 32-row tiles where they give every CU two workgroups, 8-row tiles (one wave) below
inline long long thin_tiles(const ConvArgs& a, int batch, int rows) {
  return (long long)batch * ((a.ho + rows - 1) / rows) * ((a.wo + T_TW - 1) / T_TW);
}

template <int KS, int ROWS>
int launch_rows(const ConvArgs& a, int batch, hipStream_t st) {
  const dim3 grid((a.wo + T_TW - 1) / T_TW, (a.ho + ROWS - 1) / ROWS, batch);
  const int ck = conv_chunk_of(KS, 1), slab = conv_slab_floats(KS, 1, 32);
  if (a.cout <= 2) hipLaunchKernelGGL((conv_thin_kernel<KS, 2, ROWS>), grid, dim3(8 * ROWS), 0, st, a, ck, slab);
  else if (a.cout == 3) hipLaunchKernelGGL((conv_thin_kernel<KS, 3, ROWS>), grid, dim3(8 * ROWS), 0, st, a, ck, slab);
  else hipLaunchKernelGGL((conv_thin_kernel<KS, 4, ROWS>), grid, dim3(8 * ROWS), 0, st, a, ck, slab);
  return check_launch("conv_thin_kernel");
}

template <int KS>
int launch_ks(const ConvArgs& a, int batch, hipStream_t st) {
  return thin_tiles(a, batch, 32) >= 512 ? launch_rows<KS, 32>(a, batch, st) : launch_rows<KS, 8>(a, batch, st);
}

}  // namespace

bool thin_conv_applies(const ConvArgs& a, int ksize, int stride, int batch) {
  return a.cout <= 4 && stride == 1 && (ksize == 3 || ksize == 7) && a.n_src == 1 && !a.ps && a.res_scale == nullptr &&
         a.soff_ok && thin_tiles(a, batch, 8) >= 256;   // a tile per CU at least
}

int launch_thin_conv(const ConvArgs& a, int ksize, int batch, hipStream_t st) {
  return ksize == 3 ? launch_ks<3>(a, batch, st) : launch_ks<7>(a, batch, st);
}

}  // namespace dvc
