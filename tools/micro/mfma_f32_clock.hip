// Micro-benchmark (diagnostic, not part of the library): what a CU's fp32 matrix pipe sustains on this GPU,
// and at which clock, for (a) a bare v_mfma_f32_32x32x2_f32 loop and (b) the convolution's instruction mix
// (6 ds_read_b32 per 8 MFMAs), at 1 and 2 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));

// LDSREADS: 0 = operands stay in registers, 1 = re-read from LDS per group (low-entropy values),
// 2 = re-read from LDS, full-entropy mantissas and mixed signs (the switching activity of real activations)
template <int LDSREADS>
__global__ __launch_bounds__(256, 2) void k(float* out, unsigned long long* clk, int iters) {
  __shared__ float lds[8192];
  for (int i = threadIdx.x; i < 8192; i += 256) {
    unsigned h = (unsigned)i * 2654435761u + blockIdx.x * 40503u;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    lds[i] = LDSREADS == 2 ? ((int)(h & 0xFFFFFF) - 0x800000) * (1.0f / 0x800000) : (float)((i * 7) & 31) * 0.01f;
  }
  __syncthreads();
  f32x16 acc[8];
  for (int a = 0; a < 8; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
  float av[2] = {1.0f + threadIdx.x * 1e-3f, 0.5f}, bv[4] = {0.25f, 0.5f, 0.75f, 1.f};
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  const float* base = lds + (threadIdx.x & 63);
  for (int it = 0; it < iters; ++it) {
    if (LDSREADS) {
      const int o = (it & 15) * 64;
      av[0] = base[o]; av[1] = base[o + 1024];
      bv[0] = base[o + 2048]; bv[1] = base[o + 3072]; bv[2] = base[o + 4096]; bv[3] = base[o + 5120];
    }
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int n = 0; n < 4; ++n)
        acc[m * 4 + n] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[m], bv[n], acc[m * 4 + n], 0, 0, 0);
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
  for (int a = 0; a < 8; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) { clk[blockIdx.x * 2] = t1 - t0; clk[blockIdx.x * 2 + 1] = r1 - r0; }
}

template <int L>
void run(const char* name, int blocks, int iters) {
  float* out; unsigned long long* clk;
  hipMalloc(&out, blocks * 256 * 4); hipMalloc(&clk, blocks * 16);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(k<L>, dim3(blocks), dim3(256), 0, 0, out, clk, iters);
  hipEventRecord(e0, 0);
  const int reps = 20;
  for (int w = 0; w < reps; ++w) hipLaunchKernelGGL(k<L>, dim3(blocks), dim3(256), 0, 0, out, clk, iters);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
  std::vector<unsigned long long> h(blocks * 2);
  hipMemcpy(h.data(), clk, blocks * 16, hipMemcpyDeviceToHost);
  double cyc = 0, rt = 0; for (int b = 0; b < blocks; ++b) { cyc += h[2 * b]; rt += h[2 * b + 1]; }
  const double flops = (double)blocks * 4 * iters * 8 * 2.0 * 32 * 32 * 2;
  printf("%-50s blocks %4d: %8.1f us  %6.1f TFLOP/s  clock %.3f GHz  cycles/MFMA/wave %.1f\n", name, blocks, ms * 1e3,
         flops / ms / 1e9, cyc / rt * 0.1, (cyc / blocks) / (iters * 8.0));
  hipFree(out); hipFree(clk);
}

int main() {
  const int iters = 4000;
  run<0>("bare MFMA, 1 wave/SIMD", 256, iters);
  run<0>("bare MFMA, 2 waves/SIMD", 512, iters);
  run<1>("MFMA + 6 ds_read_b32, 1 wave/SIMD", 256, iters);
  run<1>("MFMA + 6 ds_read_b32, 2 waves/SIMD", 512, iters);
  run<0>("bare MFMA, 2 waves/SIMD, 8 rounds", 4096, iters / 4);
  run<2>("MFMA + 6 ds_read_b32 of random data, 1 wave/SIMD", 256, iters);
  run<2>("MFMA + 6 ds_read_b32 of random data, 2 waves/SIMD", 512, iters);
  run<2>("same, 2 waves/SIMD, 8 rounds", 4096, iters / 4);
  return 0;
}
