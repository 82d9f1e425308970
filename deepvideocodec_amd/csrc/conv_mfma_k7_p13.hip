// Instantiates the MFMA convolution kernels for: 7x7 stride 1 precision 1 and 3 (and routes precision 2).
#include "conv_mfma_impl.h"

namespace dvc {

int conv_mfma_run_k7_p3(const ConvArgs& a, int batch, hipStream_t st) { return dispatch_tile<7, 1, 3>(a, batch, st); }

int conv_mfma_run_k7_reduced(const ConvArgs& a, int precision, int batch, hipStream_t st) {
  if (precision == DVC_PRECISION_FP16X2) return conv_mfma_run_k7_p3(a, batch, st);
  if (precision == DVC_PRECISION_BF16X2) return conv_mfma_run_k7_p2(a, batch, st);
  return dispatch_tile<7, 1, 1>(a, batch, st);
}

}  // namespace dvc
