// Direct convolution for layers with at most 4 output channels (the last layer of each SpyNet level,
// 16 -> 2 at 7x7, and the 64 -> 3 reconstruction layer): on the matrix cores such a layer keeps 2 or 3
// of 32 MFMA rows busy, here every multiply is useful.
//
// A workgroup computes an 8 x 32 tile of output pixels, one pixel per lane, all output channels in
// registers.  Input channels go through LDS eight (7x7: four) at a time as (8 + k - 1) x (32 + k - 1) halo tiles
// (two buffers: the next chunk is loaded while the current one is multiplied); the weights of a
// channel are wave-uniform and are read as scalars straight from the packed slabs the MFMA kernel uses
// ([chunk][(ci - c0) * taps + tap][32 output channels], zero filled), so the inner loop is
// one LDS read and Cout FMAs with a scalar operand per tap.  Same fused prologue / epilogue as the MFMA
// kernel for the cases these layers use: input and output LeakyReLU, bias, up to two residuals.
#include "conv_args.h"

namespace dvc {
namespace {

constexpr int T_TH = 8, T_TW = 32;
constexpr int thin_chunk(int ks) { return ks == 7 ? 4 : 8; }   // input channels per LDS buffer

template <int KS, int CO>
__global__ __launch_bounds__(256) void conv_thin_kernel(const ConvArgs a, int ck_packed, int slab) {
  constexpr int T_CK = thin_chunk(KS);
  constexpr int PAD = KS / 2, IH = T_TH + KS - 1, IW = T_TW + KS - 1, IE = IH * IW, TAPS = KS * KS;
  constexpr int SLOTS = (T_CK * IE + 255) / 256;
  __shared__ float in_s[2][T_CK * IE];
  const int tid = threadIdx.x, tx = tid & 31, ty = tid >> 5;
  const int n = blockIdx.z;
  const int oy0 = blockIdx.y * T_TH, ox0 = blockIdx.x * T_TW;
  const float* const sp = a.src[0].ptr + (long long)n * a.src[0].batch_stride;
  const long long cs = a.src[0].chan_stride;
  const int cin = a.cin;
  const float in_slope = a.in_slope;

  // element e of a chunk is (channel e / IE, halo position e % IE): the per-lane part of its address does
  // not change from chunk to chunk (32-bit element offsets: the source spans < 2 GiB, checked on the host)
  float stage[SLOTS];
  int soff[SLOTS], sch[SLOTS];   // offset inside the chunk's first channel plane + channel * stride, or -1
#pragma unroll
  for (int j = 0; j < SLOTS; ++j) {
    const int e = tid + 256 * j;
    const int c = e / IE, r = e - c * IE;
    const int iy = r / IW, ix = r - iy * IW;
    const int gy = oy0 - PAD + iy, gx = ox0 - PAD + ix;
    const bool ok = e < T_CK * IE && gy >= 0 && gy < a.h && gx >= 0 && gx < a.w;
    sch[j] = c;
    soff[j] = ok ? c * (int)cs + gy * a.w + gx : -1;
  }
  auto load_chunk = [&](int c0) {
    const float* const base = sp + (long long)c0 * cs;
#pragma unroll
    for (int j = 0; j < SLOTS; ++j) stage[j] = (soff[j] >= 0 && c0 + sch[j] < cin) ? base[soff[j]] : 0.f;
  };
  auto store_chunk = [&](int buf) {
#pragma unroll
    for (int j = 0; j < SLOTS; ++j) {
      const int e = tid + 256 * j;
      if (e < T_CK * IE) in_s[buf][e] = leaky(stage[j], in_slope);   // slope 1 = identity
    }
  };

  float acc[CO];
#pragma unroll
  for (int co = 0; co < CO; ++co) acc[co] = 0.f;

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
      const float* xin = in_s[cur] + c * IE + ty * IW + tx;
#pragma unroll
      for (int ky = 0; ky < KS; ++ky)
#pragma unroll
        for (int kx = 0; kx < KS; ++kx) {
          const float v = xin[ky * IW + kx];
#pragma unroll
          for (int co = 0; co < CO; ++co) acc[co] = fmaf(wrow[(ky * KS + kx) * 32 + co], v, acc[co]);
        }
    }
    if (k + 1 < nchunks) store_chunk(cur ^ 1);   // that buffer was last read before the previous barrier
    __syncthreads();
  }

  const int oy = oy0 + ty, ox = ox0 + tx;
  if (oy >= a.ho || ox >= a.wo) return;
  const long long pix = (long long)oy * a.wo + ox;
#pragma unroll
  for (int co = 0; co < CO; ++co) {
    if (co >= a.cout) break;
    float v = acc[co] + (a.bias != nullptr ? a.bias[co] : 0.f);
    v = leaky(v, a.out_slope);
    if (a.res != nullptr) {
      v = __fadd_rn(v, a.res[(long long)n * a.rbs + co * a.rcs + pix]);
      if (a.res2 != nullptr) v = __fadd_rn(v, a.res2[(long long)n * a.r2bs + co * a.r2cs + pix]);
    }
    a.out[(long long)n * a.obs + co * a.ocs + pix] = v;
  }
}

template <int KS>
int launch_ks(const ConvArgs& a, int batch, hipStream_t st) {
  const dim3 grid((a.wo + T_TW - 1) / T_TW, (a.ho + T_TH - 1) / T_TH, batch);
  const int ck = conv_chunk_of(KS, 1), slab = conv_slab_floats(KS, 1, 32);
  if (a.cout <= 2) hipLaunchKernelGGL((conv_thin_kernel<KS, 2>), grid, dim3(256), 0, st, a, ck, slab);
  else if (a.cout == 3) hipLaunchKernelGGL((conv_thin_kernel<KS, 3>), grid, dim3(256), 0, st, a, ck, slab);
  else hipLaunchKernelGGL((conv_thin_kernel<KS, 4>), grid, dim3(256), 0, st, a, ck, slab);
  return check_launch("conv_thin_kernel");
}

}  // namespace

bool thin_conv_applies(const ConvArgs& a, int ksize, int stride, int batch) {
  return a.cout <= 4 && stride == 1 && (ksize == 3 || ksize == 7) && a.n_src == 1 && !a.ps && a.res_scale == nullptr &&
         a.soff_ok &&
         (long long)batch * ((a.ho + T_TH - 1) / T_TH) * ((a.wo + T_TW - 1) / T_TW) >= 256;   // a tile per CU at least
}

int launch_thin_conv(const ConvArgs& a, int ksize, int batch, hipStream_t st) {
  return ksize == 3 ? launch_ks<3>(a, batch, st) : launch_ks<7>(a, batch, st);
}

}  // namespace dvc
