// Instantiates the MFMA convolution kernels for: 3x3 stride 1 precision 0.
#include "conv_mfma_impl.h"

namespace dvc {

int conv_mfma_run_f32_k3s1(const ConvArgs& a, int ksize, int stride, int batch, hipStream_t st) {
  return dispatch_tile<3, 1, 0>(a, batch, st);
}

}  // namespace dvc

#ifdef DVC_CONV_STAMPS
// diagnostic build: point the 3x3 stride-1 kernels' phase stamps at a caller buffer of 16 x workgroups 64-bit words
extern "C" int dvc_debug_conv_stamps(unsigned long long* buf) {
  return hipMemcpyToSymbol(HIP_SYMBOL(g_conv_stamps), &buf, sizeof(buf)) == hipSuccess ? 0 : -2;
}
#endif
