// Instantiates the MFMA convolution kernels for: 1x1 stride 1 precision 0, 1x1 stride 2 precision 0.
#include "conv_mfma_impl.h"

namespace dvc {

int conv_mfma_run_f32_k1(const ConvArgs& a, int ksize, int stride, int batch, hipStream_t st) {
  if (ksize == 1 && stride == 1) return dispatch_tile<1, 1, 0>(a, batch, st);
  return dispatch_tile<1, 2, 0>(a, batch, st);
}

}  // namespace dvc
