// Instantiates the MFMA convolution kernels for: 3x3 stride 1 precision 2, 3x3 stride 2 precision 2.
#include "conv_mfma_impl.h"

namespace dvc {

int conv_mfma_run_p2_k3(const ConvArgs& a, int ksize, int stride, int batch, hipStream_t st) {
  if (ksize == 3 && stride == 1) return dispatch_tile<3, 1, 2>(a, batch, st);
  return dispatch_tile<3, 2, 2>(a, batch, st);
}

}  // namespace dvc
