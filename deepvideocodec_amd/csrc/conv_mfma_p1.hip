// Instantiates the MFMA convolution kernels for: 3x3 stride 1 precision 1, 3x3 stride 2 precision 1, 1x1 stride 1 precision 1, 1x1 stride 2 precision 1.
#include "conv_mfma_impl.h"

namespace dvc {

int conv_mfma_run_p1(const ConvArgs& a, int ksize, int stride, int batch, hipStream_t st) {
  if (ksize == 3 && stride == 1) return dispatch_tile<3, 1, 1>(a, batch, st);
  if (ksize == 3 && stride == 2) return dispatch_tile<3, 2, 1>(a, batch, st);
  if (ksize == 1 && stride == 1) return dispatch_tile<1, 1, 1>(a, batch, st);
  return dispatch_tile<1, 2, 1>(a, batch, st);
}

}  // namespace dvc
