// Instantiates the MFMA convolution kernels for: 3x3 stride 2 precision 0.
#include "conv_mfma_impl.h"

namespace dvc {

int conv_mfma_run_f32_k3s2(const ConvArgs& a, int ksize, int stride, int batch, hipStream_t st) {
  return dispatch_tile<3, 2, 0>(a, batch, st);
}

}  // namespace dvc
