// Launch arguments shared by the convolution kernels (conv_mfma.hip, conv_thin.hip).
#pragma once
#include <hip/hip_runtime.h>

#include "dvc_common.h"

namespace dvc {

struct ConvArgs {
  DvcSrc src[DVC_MAX_SRC];
  int n_src;
  int cin, h, w;         // input
  int cout, ho, wo;      // conv output (before shuffle)
  const float* wpacked;  // [cout block][chunk][slab], see pack_weight_kernel
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
  int vec4;               // output / residual rows allow 16-byte accesses
  int soff_ok;            // every source spans < 2 GiB per sample: a channel can be a 32-bit scalar offset
  float* plane_sums;      // optional [batch][cout][sum_slots]: per-workgroup sums of the stored output planes
  int sum_slots;          // workgroups per (sample, channel block) of this launch
  // persistent (wave-specialised) launches: the grid is one workgroup per CU and each walks a range of the launch's
  // work items, numbered (sample, channel block, tile) with the tile fastest
  int tiles_xy, cblocks, total_items;
};

// fp32 packed weights (dvc_conv2d_pack_weight): input channels per K-chunk and floats per slab
constexpr int conv_chunk_of(int ks, int s) { return ks == 7 ? 2 : ks == 1 ? 16 : (s == 2 ? 4 : 8); }
constexpr int conv_slab_floats(int ks, int s, int cb) { return (conv_chunk_of(ks, s) * ks * ks * cb + 255) / 256 * 256; }

// Row-pair image of a 7x7 stride-1 layer with exactly 16 output channels (appended to the ordinary image by
// dvc_conv2d_pack_weight): a 32-row block whose rows 0..15 are the filters and rows 16..31 the same filters moved
// one tap row down, over 8 x 7 taps -- one pass of the 32-row matrix instruction then yields two output rows.
constexpr int CONV_ROWPAIR_COUT = 16, CONV_ROWPAIR_TAPS = 8 * 7;
constexpr bool conv_has_rowpair_image(int cout, int ks, int s) { return ks == 7 && s == 1 && cout == CONV_ROWPAIR_COUT; }
constexpr int conv_rowpair_slab_floats() { return (conv_chunk_of(7, 1) * CONV_ROWPAIR_TAPS * 32 + 255) / 256 * 256; }

// conv_mfma_*.hip: the MFMA kernel variants, split over translation units by shape and precision
// (ksize / stride select inside a unit; the caller has validated them)
int conv_mfma_run_f32_k3s1(const ConvArgs& a, int ksize, int stride, int batch, hipStream_t st);
int conv_mfma_run_f32_k3s2(const ConvArgs& a, int ksize, int stride, int batch, hipStream_t st);
int conv_mfma_run_f32_k1(const ConvArgs& a, int ksize, int stride, int batch, hipStream_t st);
int conv_mfma_run_f32_k7(const ConvArgs& a, int ksize, int stride, int batch, hipStream_t st);
int conv_mfma_run_p1(const ConvArgs& a, int ksize, int stride, int batch, hipStream_t st);
int conv_mfma_run_p2_k3(const ConvArgs& a, int ksize, int stride, int batch, hipStream_t st);
int conv_mfma_run_p2_k1(const ConvArgs& a, int ksize, int stride, int batch, hipStream_t st);
int conv_mfma_run_p3_k3(const ConvArgs& a, int ksize, int stride, int batch, hipStream_t st);
int conv_mfma_run_p3_k1(const ConvArgs& a, int ksize, int stride, int batch, hipStream_t st);

// conv_mfma_k7_p*.hip: 7x7 stride 1 in the reduced-operand modes (precision 1..3)
int conv_mfma_run_k7_reduced(const ConvArgs& a, int precision, int batch, hipStream_t st);
int conv_mfma_run_k7_p2(const ConvArgs& a, int batch, hipStream_t st);
int conv_mfma_run_k7_p3(const ConvArgs& a, int batch, hipStream_t st);

// conv_thin.hip: direct convolution for layers with at most 4 output channels
bool thin_conv_applies(const ConvArgs& a, int ksize, int stride, int batch);
int launch_thin_conv(const ConvArgs& a, int ksize, int batch, hipStream_t st);

}  // namespace dvc
