// Instantiates the MFMA convolution kernels for: 7x7 stride 1 precision 0.
#include "conv_mfma_impl.h"

namespace {

// 7x7 stride 1 with exactly 16 output channels (the fourth layer of every SpyNet level: half the MFMA rows would
// idle): the row-pair form, when the launch is one it covers -- whole tiles, 16-byte rows, plain epilogue -- with
// the tallest tile that divides the height and still yields two workgroups per CU.  `rowpair_image`: where the
// packed row-pair image starts (behind the ordinary one).  Returns -1 when the ordinary form has to run.
inline int launch_rowpair(const ConvArgs& a, int batch, hipStream_t st) {
  if (!dvc::conv_has_rowpair_image(a.cout, 7, 1) || !a.vec4 || a.res != nullptr || a.ps || a.wo % 32 != 0 ||
      a.out_slope < 0.f || a.out_slope > 1.f)
    return -1;
  const int nchunks = (a.cin + chunk_of(7, 1) - 1) / chunk_of(7, 1);
  ConvArgs b = a;
  b.wpacked = a.wpacked + (long long)nchunks * slab_floats(7, 1, 32);
  const long long tx = a.wo / 32;
  for (int nt = 4; nt >= 1; nt >>= 1) {
    if (a.ho % (8 * nt) != 0) continue;
    const long long blocks = (long long)(a.ho / (8 * nt)) * tx * batch;
    if (blocks < 480 && nt > 1) continue;
    if (nt == 4) return launch<7, 1, 1, 4, false, 0, 2>(b, batch, st);
    if (nt == 2) return launch<7, 1, 1, 2, false, 0, 2>(b, batch, st);
    return launch<7, 1, 1, 1, false, 0, 2>(b, batch, st);
  }
  return -1;
}

}  // namespace

namespace dvc {

int conv_mfma_run_f32_k7(const ConvArgs& a, int ksize, int stride, int batch, hipStream_t st) {
  const int rc = launch_rowpair(a, batch, st);
  if (rc != -1) return rc;
  return dispatch_tile<7, 1, 0>(a, batch, st);
}

}  // namespace dvc
