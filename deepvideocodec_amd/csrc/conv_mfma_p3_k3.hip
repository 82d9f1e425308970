// Instantiates the MFMA convolution kernels for: 3x3 stride 1 precision 3, 3x3 stride 2 precision 3.
#include "conv_mfma_impl.h"

namespace dvc {

int conv_mfma_run_p3_k3(const ConvArgs& a, int ksize, int stride, int batch, hipStream_t st) {
  if (ksize == 3 && stride == 1) return dispatch_tile<3, 1, 3>(a, batch, st);
  return dispatch_tile<3, 2, 3>(a, batch, st);
}

}  // namespace dvc
