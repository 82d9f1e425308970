// Instantiates the MFMA convolution kernels for: 7x7 stride 1 precision 2.
#include "conv_mfma_impl.h"

namespace dvc {

int conv_mfma_run_k7_p2(const ConvArgs& a, int batch, hipStream_t st) { return dispatch_tile<7, 1, 2>(a, batch, st); }

}  // namespace dvc
