// Instantiates the MFMA convolution kernels for: 1x1 stride 1 precision 3, 1x1 stride 2 precision 3.
#include "conv_mfma_impl.h"

namespace dvc {

int conv_mfma_run_p3_k1(const ConvArgs& a, int ksize, int stride, int batch, hipStream_t st) {
  if (ksize == 1 && stride == 1) return dispatch_tile<1, 1, 3>(a, batch, st);
  return dispatch_tile<1, 2, 3>(a, batch, st);
}

}  // namespace dvc
