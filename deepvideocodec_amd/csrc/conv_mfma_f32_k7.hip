// Instantiates the MFMA convolution kernels for: 7x7 stride 1 precision 0.
#include "conv_mfma_impl.h"

namespace dvc {

int conv_mfma_run_f32_k7(const ConvArgs& a, int ksize, int stride, int batch, hipStream_t st) {
  const int rc = launch_rowpair(a, batch, st);
  if (rc != -1) return rc;
  return dispatch_tile<7, 1, 0>(a, batch, st);
}

}  // namespace dvc
