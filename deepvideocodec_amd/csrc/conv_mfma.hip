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
// Pipeline.  The K loop walks the input channels in chunks of CK, with two LDS
// buffers for the input tile and for the weight slab.  While the MFMAs of chunk k
// run, chunk k+1 is in flight: its weight slab goes global -> LDS directly
// (global_load_lds, 16 B per lane, no registers; weights are pre-packed so a slab
// is one contiguous range), its input halo tile goes into registers and is written
// to the other LDS buffer after the MFMA phase.  One barrier per chunk.
//
// Fused around the MFMA loop: channel concat of up to 4 sources and LeakyReLU on
// the input while staging; bias, LeakyReLU/ReLU, (scaled) residual adds and the
// PixelShuffle(2) scatter in the epilogue.
#include "conv_mfma_impl.h"

namespace {

template <bool F16>
__global__ __launch_bounds__(256) void pack_weight_half_kernel(const float* w, typename Half16<F16>::T* packed, int cout, int cin, int ks,
                                                               int cb, int part_elems, int parts, int nchunks,
                                                               long long total) {
  const int taps = ks * ks;
  const int slab_elems = part_elems * parts;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int e = (int)(i % slab_elems);
    const long long t = i / slab_elems;
    const int chunk = (int)(t % nchunks), blk = (int)(t / nchunks);
    const int part = e / part_elems, ep = e - part * part_elems;
    const int e8 = ep & 7, u = ep >> 3;
    const int col = u % cb, q = u / cb;
    const int o = q & 1, tap = q >> 1;
    const int ci = chunk * BF_CK + o * 8 + e8, co = blk * cb + col;
    float v = 0.f;
    if (tap < taps && ci < cin && co < cout) v = w[((long long)co * cin + ci) * taps + tap];
    const auto hi = to_half16<F16>(v);
    packed[i] = part == 0 ? hi : to_half16<F16>(v - (float)hi);
  }
}

// Reduced-operand image of a 3x3 / 7x7 stride-1 layer: [cout block][chunk of 16 channels][tap row ky][step slab], a
// step slab = [part][kx][octet][co][8 channels] in 16-bit elements, zero filled past Cin / Cout / the slab's payload.
template <bool F16>
__global__ __launch_bounds__(256) void pack_weight_step_kernel(const float* w, typename Half16<F16>::T* packed, int cout, int cin,
                                                               int ks, int cb, int parts, int step_elems, int nchunks,
                                                               long long total) {
  const int payload = parts * ks * 2 * cb * 8;   // 16-bit elements of a slab that carry weights
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int e = (int)(i % step_elems);
    long long t = i / step_elems;
    const int ky = (int)(t % ks);
    t /= ks;
    const int chunk = (int)(t % nchunks), blk = (int)(t / nchunks);
    float v = 0.f;
    int part = 0;
    if (e < payload) {
      const int e8 = e & 7;
      int u = e >> 3;
      const int col = u % cb;
      u /= cb;
      const int o = u & 1;
      u >>= 1;
      const int kx = u % ks;
      part = u / ks;
      const int ci = chunk * BF_CK + o * 8 + e8, co = blk * cb + col;
      if (ci < cin && co < cout) v = w[((long long)co * cin + ci) * ks * ks + ky * ks + kx];
    }
    const auto hi = to_half16<F16>(v);
    packed[i] = part == 0 ? hi : to_half16<F16>(v - (float)hi);
  }
}

// OIHW -> [cout block][chunk][row = (ci - c0) * taps + tap][co - cb0], zero filled past Cin / Cout.
// `dgrad`: the image of the data-gradient convolution's weight, W'[co][ci][ky][kx] = W[ci][co][k-1-ky][k-1-kx]
// (cout / cin here are those of W': the input / output channels of the layer), read straight from the layer's OIHW
// tensor -- no flipped / transposed copy is made.
__global__ __launch_bounds__(256) void pack_weight_kernel(const float* w, float* packed, int cout, int cin, int ks,
                                                          int ck, int cb, int slab, int nchunks, long long total, int dgrad) {
  const int taps = ks * ks;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int e = (int)(i % slab);
    const long long t = i / slab;
    const int chunk = (int)(t % nchunks), blk = (int)(t / nchunks);
    const int row = e / cb, col = e - row * cb;
    const int ci = chunk * ck + row / taps, tap = row % taps;
    const int co = blk * cb + col;
    float v = 0.f;
    if (row < ck * taps && ci < cin && co < cout)
      v = dgrad ? w[((long long)ci * cout + co) * taps + (taps - 1 - tap)] : w[((long long)co * cin + ci) * taps + tap];
    packed[i] = v;
  }
}

// Row-pair image of a 7x7 layer with 16 output channels: [chunk][row = (ci - c0) * 56 + ky' * 7 + kx][32 columns],
// column dy * 16 + co = w[co][ci][ky' - dy][kx] where that tap row exists, else 0.
__global__ __launch_bounds__(256) void pack_weight_rowpair_kernel(const float* w, float* packed, int cin, int ck, int slab,
                                                                  int nchunks, long long total, int dgrad) {
  constexpr int taps = dvc::CONV_ROWPAIR_TAPS;
  static_assert(dvc::CONV_ROWPAIR_COUT == 16, "column = dy * 16 + co");
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int e = (int)(i % slab), chunk = (int)(i / slab);
    const int row = e / 32, col = e - row * 32;
    const int ci = chunk * ck + row / taps, tap = row % taps;
    const int kyp = tap / 7, kx = tap - kyp * 7;
    const int dy = col >> 4, co = col & 15, ky = kyp - dy;
    float v = 0.f;
    if (row < ck * taps && ci < cin && ky >= 0 && ky < 7)
      v = dgrad ? w[((long long)ci * 16 + co) * 49 + 48 - (ky * 7 + kx)] : w[((long long)co * cin + ci) * 49 + ky * 7 + kx];
    packed[i] = v;
  }
}

int check_geometry(int cout, int cin, int ks, int stride, const char* who) {
  DVC_REQUIRE(ks == 1 || ks == 3 || ks == 7, "%s: ksize=%d unsupported", who, ks);
  DVC_REQUIRE(stride == 1 || stride == 2, "%s: stride=%d unsupported", who, stride);
  DVC_REQUIRE(!(ks == 7 && stride != 1), "%s: 7x7 stride 2 unsupported", who);
  DVC_REQUIRE(cout >= 1 && cin >= 1, "%s: bad channel counts", who);
  DVC_REQUIRE((long long)cout * cin * ks * ks < (1LL << 31), "%s: weight too large", who);
  return DVC_OK;
}

}  // namespace

extern "C" int64_t dvc_conv2d_packed_weight_floats(int32_t cout, int32_t cin, int32_t ksize, int32_t stride) {
  if (check_geometry(cout, cin, ksize, stride, "dvc_conv2d_packed_weight_floats")) return DVC_EINVAL;
  const int cb = cout_block_of(cout), ck = chunk_of(ksize, stride);
  const int64_t nchunks = (cin + ck - 1) / ck;
  int64_t total = (int64_t)((cout + cb - 1) / cb) * nchunks * slab_floats(ksize, stride, cb);
  if (dvc::conv_has_rowpair_image(cout, ksize, stride)) total += nchunks * dvc::conv_rowpair_slab_floats();
  return total;
}

namespace {
int pack_weight_impl(const float* weight, int32_t cout, int32_t cin, int32_t ksize, int32_t stride, float* packed, void* stream,
                     int dgrad);
}
extern "C" int dvc_conv2d_pack_weight(const float* weight, int32_t cout, int32_t cin, int32_t ksize, int32_t stride,
                                      float* packed, void* stream) {
  return pack_weight_impl(weight, cout, cin, ksize, stride, packed, stream, 0);
}
extern "C" int dvc_conv2d_pack_weight_dgrad(const float* weight, int32_t cout, int32_t cin, int32_t ksize, float* packed,
                                            void* stream) {
  // the data gradient of a layer with `cout` output and `cin` input channels is a stride-1 convolution with cin
  // output and cout input channels
  return pack_weight_impl(weight, cin, cout, ksize, 1, packed, stream, 1);
}
namespace {
int pack_weight_impl(const float* weight, int32_t cout, int32_t cin, int32_t ksize, int32_t stride, float* packed, void* stream,
                     int dgrad) {
  if (int rc = check_geometry(cout, cin, ksize, stride, "dvc_conv2d_pack_weight")) return rc;
  DVC_REQUIRE(weight != nullptr && packed != nullptr, "dvc_conv2d_pack_weight: null pointer");
  DVC_REQUIRE(reinterpret_cast<uintptr_t>(packed) % 16 == 0, "dvc_conv2d_pack_weight: packed buffer must be 16-byte aligned");
  const int cb = cout_block_of(cout), ck = chunk_of(ksize, stride);
  const int slab = slab_floats(ksize, stride, cb), nchunks = (cin + ck - 1) / ck;
  const long long total = (long long)((cout + cb - 1) / cb) * nchunks * slab;
  hipLaunchKernelGGL(pack_weight_kernel, dim3(dvc::stream_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, weight,
                     packed, cout, cin, ksize, ck, cb, slab, nchunks, total, dgrad);
  if (int rc = dvc::check_launch("pack_weight_kernel")) return rc;
  if (dvc::conv_has_rowpair_image(cout, ksize, stride)) {   // second image behind the first (see launch_rowpair)
    const int rslab = dvc::conv_rowpair_slab_floats();
    const long long rtotal = (long long)nchunks * rslab;
    hipLaunchKernelGGL(pack_weight_rowpair_kernel, dim3(dvc::stream_grid(rtotal, 256)), dim3(256), 0, (hipStream_t)stream,
                       weight, packed + total, cin, ck, rslab, nchunks, rtotal, dgrad);
    return dvc::check_launch("pack_weight_rowpair_kernel");
  }
  return DVC_OK;
}
}  // namespace

extern "C" int64_t dvc_conv2d_reduced_packed_weight_floats(int32_t cout, int32_t cin, int32_t ksize, int32_t stride,
                                                           int32_t precision) {
  if (check_geometry(cout, cin, ksize, stride, "dvc_conv2d_reduced_packed_weight_floats")) return DVC_EINVAL;
  if ((ksize == 7 && stride != 1) || precision < DVC_PRECISION_BF16 || precision > DVC_PRECISION_FP16X2) {
    dvc::set_error("dvc_conv2d_reduced_packed_weight_floats: precision 1..3");
    return DVC_EINVAL;
  }
  const int cb = cout_block_of(cout), parts = precision == DVC_PRECISION_BF16 ? 1 : 2;
  const int64_t slabs = (int64_t)((cout + cb - 1) / cb) * ((cin + BF_CK - 1) / BF_CK);
  if (reduced_stepped(ksize, stride)) return slabs * ksize * step_slab_floats(ksize, cb, parts);
  return slabs * parts * slab_floats_bf16(ksize, cb);
}

extern "C" int dvc_conv2d_reduced_pack_weight(const float* weight, int32_t cout, int32_t cin, int32_t ksize,
                                              int32_t stride, int32_t precision, float* packed, void* stream) {
  if (int rc = check_geometry(cout, cin, ksize, stride, "dvc_conv2d_reduced_pack_weight")) return rc;
  DVC_REQUIRE(precision >= DVC_PRECISION_BF16 && precision <= DVC_PRECISION_FP16X2,
              "dvc_conv2d_reduced_pack_weight: precision 1..3");
  DVC_REQUIRE(weight != nullptr && packed != nullptr, "dvc_conv2d_reduced_pack_weight: null pointer");
  DVC_REQUIRE(reinterpret_cast<uintptr_t>(packed) % 16 == 0, "dvc_conv2d_reduced_pack_weight: packed buffer must be 16-byte aligned");
  const int cb = cout_block_of(cout), parts = precision == DVC_PRECISION_BF16 ? 1 : 2;
  if (reduced_stepped(ksize, stride)) {
    const int step_elems = 2 * step_slab_floats(ksize, cb, parts), nch = (cin + BF_CK - 1) / BF_CK;
    const long long tot = (long long)((cout + cb - 1) / cb) * nch * ksize * step_elems;
    const dim3 g(dvc::stream_grid(tot, 256));
    if (precision == DVC_PRECISION_FP16X2)
      hipLaunchKernelGGL(pack_weight_step_kernel<true>, g, dim3(256), 0, (hipStream_t)stream, weight,
                         reinterpret_cast<_Float16*>(packed), cout, cin, ksize, cb, parts, step_elems, nch, tot);
    else
      hipLaunchKernelGGL(pack_weight_step_kernel<false>, g, dim3(256), 0, (hipStream_t)stream, weight,
                         reinterpret_cast<__bf16*>(packed), cout, cin, ksize, cb, parts, step_elems, nch, tot);
    return dvc::check_launch("pack_weight_step_kernel");
  }
  const int part_elems = 2 * slab_floats_bf16(ksize, cb), nchunks = (cin + BF_CK - 1) / BF_CK;
  const long long total = (long long)((cout + cb - 1) / cb) * nchunks * part_elems * parts;
  const dim3 grid(dvc::stream_grid(total, 256));
  if (precision == DVC_PRECISION_FP16X2)
    hipLaunchKernelGGL(pack_weight_half_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, weight,
                       reinterpret_cast<_Float16*>(packed), cout, cin, ksize, cb, part_elems, parts, nchunks, total);
  else
    hipLaunchKernelGGL(pack_weight_half_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, weight,
                       reinterpret_cast<__bf16*>(packed), cout, cin, ksize, cb, part_elems, parts, nchunks, total);
  return dvc::check_launch("pack_weight_half_kernel");
}

namespace {
// Validates the parameters and fills the launch arguments (everything but the tile choice).
int build_conv_args(const DvcConv2d* p, ConvArgs& a) {
  DVC_REQUIRE(p != nullptr, "dvc_conv2d: null params");
  DVC_REQUIRE(p->n_src >= 1 && p->n_src <= DVC_MAX_SRC, "dvc_conv2d: n_src=%d not in 1..%d", p->n_src, DVC_MAX_SRC);
  DVC_REQUIRE(p->batch >= 1 && p->in_h >= 1 && p->in_w >= 1, "dvc_conv2d: bad dims");
  DVC_REQUIRE(p->batch <= 65535, "dvc_conv2d: batch too large");
  DVC_REQUIRE((long long)p->in_h * p->in_w < (1LL << 28), "dvc_conv2d: plane too large");
  DVC_REQUIRE(p->weight_packed != nullptr && p->out.ptr != nullptr, "dvc_conv2d: null weight/out");
  DVC_REQUIRE(reinterpret_cast<uintptr_t>(p->weight_packed) % 16 == 0, "dvc_conv2d: packed weight must be 16-byte aligned");
  DVC_REQUIRE(!p->pixel_shuffle || p->out_channels % 4 == 0, "dvc_conv2d: pixel_shuffle needs Cout %% 4 == 0");
  a = ConvArgs{};
  int cin = 0;
  for (int s = 0; s < p->n_src; ++s) {
    DVC_REQUIRE(p->src[s].ptr != nullptr && p->src[s].channels >= 1, "dvc_conv2d: bad source %d", s);
    DVC_REQUIRE(p->src[s].chan_stride >= (int64_t)p->in_h * p->in_w, "dvc_conv2d: source %d channel stride < H*W", s);
    a.src[s] = p->src[s];
    cin += p->src[s].channels;
  }
  if (int rc = check_geometry(p->out_channels, cin, p->ksize, p->stride, "dvc_conv2d")) return rc;
  a.n_src = p->n_src;
  a.cin = cin;
  a.h = p->in_h;
  a.w = p->in_w;
  a.cout = p->out_channels;
  const int pad = p->ksize / 2;
  a.ho = (p->in_h + 2 * pad - p->ksize) / p->stride + 1;
  a.wo = (p->in_w + 2 * pad - p->ksize) / p->stride + 1;
  a.wpacked = p->weight_packed;
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
  DVC_REQUIRE(!a.res_scale || a.res, "dvc_conv2d: res_scale needs residual");
  // per-lane offsets inside a 4-channel group are 32-bit
  DVC_REQUIRE(4 * a.ocs + plane < (1LL << 30) && 4 * a.rcs + plane < (1LL << 30) && 4 * a.r2cs + plane < (1LL << 30),
              "dvc_conv2d: channel stride too large for 32-bit lane offsets");
  if (a.ps) {
    DVC_REQUIRE(reinterpret_cast<uintptr_t>(a.out) % 8 == 0 && a.obs % 2 == 0 && a.ocs % 2 == 0,
                "dvc_conv2d: pixel_shuffle output must be 8-byte aligned with even strides");
    DVC_REQUIRE(!a.res || (reinterpret_cast<uintptr_t>(a.res) % 8 == 0 && a.rbs % 2 == 0 && a.rcs % 2 == 0),
                "dvc_conv2d: pixel_shuffle residual must be 8-byte aligned with even strides");
    DVC_REQUIRE(!a.res2 || (reinterpret_cast<uintptr_t>(a.res2) % 8 == 0 && a.r2bs % 2 == 0 && a.r2cs % 2 == 0),
                "dvc_conv2d: pixel_shuffle residual2 must be 8-byte aligned with even strides");
  }
  DVC_REQUIRE((long long)((a.ho + 3) / 4) * ((a.wo + 31) / 32) < 2147483647LL, "dvc_conv2d: too many tiles");
  {
    // 16-byte epilogue accesses: every row start and every 4-pixel (shuffle: 2-pixel) group aligned
    auto al = [](const void* q, long long bs, long long cs) {
      return reinterpret_cast<uintptr_t>(q) % 16 == 0 && bs % 4 == 0 && cs % 4 == 0;
    };
    // (a shuffled store covers conv pixels 2k, 2k+1; a plain one pixels 4k .. 4k+3)
    a.vec4 = (a.ps ? a.wo % 2 == 0 : a.wo % 4 == 0) && al(a.out, a.obs, a.ocs) && (!a.res || al(a.res, a.rbs, a.rcs)) &&
             (!a.res2 || al(a.res2, a.r2bs, a.r2cs));
  }
  a.soff_ok = 1;
  for (int i = 0; i < p->n_src; ++i)
    if (((long long)p->src[i].channels + 1) * p->src[i].chan_stride * 4 >= (1LL << 31)) a.soff_ok = 0;
  a.in_slope = p->in_act ? p->in_slope : 1.0f;    // leaky(v, 1) == v exactly
  a.out_slope = p->out_act ? p->out_slope : 1.0f;
  a.in_act = p->in_act ? 1 : 0;
  a.out_act = p->out_act ? 1 : 0;
  a.plane_sums = p->plane_sums;
  return DVC_OK;
}
}  // namespace

extern "C" int32_t dvc_conv2d_plane_sum_slots(const DvcConv2d* p) {
  ConvArgs a;
  if (int rc = build_conv_args(p, a)) return rc;
  return conv_plane_sum_slots(a, p->ksize, p->stride, p->precision, p->batch);
}

extern "C" int dvc_conv2d(const DvcConv2d* p, void* stream) {
  ConvArgs a;
  if (int rc = build_conv_args(p, a)) return rc;
  DVC_REQUIRE(a.plane_sums == nullptr || conv_plane_sum_slots(a, p->ksize, p->stride, p->precision, p->batch) > 0,
              "dvc_conv2d: plane_sums is not available for this launch (dvc_conv2d_plane_sum_slots() == 0)");
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (p->ksize == 1 && p->stride == 1 && !a.ps && ((long long)a.h * a.w) % 32 == 0) {
    // A 1x1 stride-1 convolution does not see the image's rows: run it on the plane as rows of 32 pixels, so that
    // a tile (4 NT rows of 32) is one contiguous run of every channel plane instead of 4 NT pieces of 128 bytes a
    // row pitch apart.  Same products in the same order for every pixel.
    a.h = a.ho = (int)(((long long)a.h * a.w) / 32);
    a.w = a.wo = 32;
    auto al = [](const void* q, long long bs, long long cs) {
      return reinterpret_cast<uintptr_t>(q) % 16 == 0 && bs % 4 == 0 && cs % 4 == 0;
    };
    a.vec4 = al(a.out, a.obs, a.ocs) && (!a.res || al(a.res, a.rbs, a.rcs)) && (!a.res2 || al(a.res2, a.r2bs, a.r2cs));
  }
  if (p->precision != DVC_PRECISION_FP32) {
    DVC_REQUIRE(p->precision >= DVC_PRECISION_BF16 && p->precision <= DVC_PRECISION_FP16X2,
                "dvc_conv2d: precision %d unknown", p->precision);
    if (p->ksize == 7) return dvc::conv_mfma_run_k7_reduced(a, p->precision, p->batch, st);
    if (p->precision == DVC_PRECISION_BF16) return dvc::conv_mfma_run_p1(a, p->ksize, p->stride, p->batch, st);
    if (p->precision == DVC_PRECISION_BF16X2)
      return p->ksize == 3 ? dvc::conv_mfma_run_p2_k3(a, 3, p->stride, p->batch, st)
                           : dvc::conv_mfma_run_p2_k1(a, 1, p->stride, p->batch, st);
    return p->ksize == 3 ? dvc::conv_mfma_run_p3_k3(a, 3, p->stride, p->batch, st)
                         : dvc::conv_mfma_run_p3_k1(a, 1, p->stride, p->batch, st);
  }
  // a handful of output channels would leave 28+ of the 32 MFMA rows idle: plain FMAs on the vector ALUs
  if (a.plane_sums == nullptr && dvc::thin_conv_applies(a, p->ksize, p->stride, p->batch))
    return dvc::launch_thin_conv(a, p->ksize, p->batch, st);
  if (p->ksize == 3)
    return p->stride == 1 ? dvc::conv_mfma_run_f32_k3s1(a, 3, 1, p->batch, st) : dvc::conv_mfma_run_f32_k3s2(a, 3, 2, p->batch, st);
  if (p->ksize == 1) return dvc::conv_mfma_run_f32_k1(a, 1, p->stride, p->batch, st);
  return dvc::conv_mfma_run_f32_k7(a, 7, 1, p->batch, st);
}
