// Instantiates the MFMA convolution kernels for: 1x1 stride 1 precision 2, 1x1 stride 2 precision 2.
#include "conv_mfma_impl.h"

namespace dvc {

int conv_mfma_run_p2_k1(const ConvArgs& a, int ksize, int stride, int batch, hipStream_t st) {
  if (ksize == 1 && stride == 1) return dispatch_tile<1, 1, 2>(a, batch, st);
  return dispatch_tile<1, 2, 2>(a, batch, st);
}

}  // namespace dvc
