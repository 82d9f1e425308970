// Instantiates the MFMA convolution kernels for: 3x3 stride 1 precision 3, 3x3 stride 2 precision 3.
#include "conv_mfma_impl.h"

namespace dvc {

int conv_mfma_run_p3_k3(const ConvArgs& a, int ksize, int stride, int batch, hipStream_t st) {
  if (ksize == 3 && stride == 1) return dispatch_tile<3, 1, 3>(a, batch, st);
  return dispatch_tile<3, 2, 3>(a, batch, st);
}

}  // namespace dvc

#ifdef DVC_CONV_STAMPS
// diagnostic build (make stamps_p3): point this unit's phase stamps at a caller buffer of 16 x workgroups 64-bit words
extern "C" int dvc_debug_conv_stamps(unsigned long long* buf) {
  return hipMemcpyToSymbol(HIP_SYMBOL(g_conv_stamps), &buf, sizeof(buf)) == hipSuccess ? 0 : -2;
}
#endif
