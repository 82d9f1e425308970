// Micro-benchmark (diagnostic): the convolution's inner loop -- per tap 2 A reads + 4 B reads (ds_read_b32)
// feeding 8 v_mfma_f32_32x32x2_f32 -- software-pipelined one tap ahead, (a) as the compiler schedules it and
// (b) with the LDS reads as inline asm and hand-counted s_waitcnt lgkmcnt(N).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float lds_read(const float* p, int off_bytes_dummy) { return *p; }

template <int MODE>
__global__ __launch_bounds__(256, 2) void k(float* out, unsigned long long* clk, int iters) {
  __shared__ float lds[16384];
  for (int i = threadIdx.x; i < 16384; i += 256) lds[i] = (float)((i * 7) & 31) * 0.01f;
  __syncthreads();
  f32x16 acc[8];
  for (int a = 0; a < 8; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  const float* base = lds + (threadIdx.x & 63);
  const unsigned lbase = (unsigned)(size_t)(__attribute__((address_space(3))) float*)base;
  float av[2][2], bv[2][4];
  constexpr int TAPS = 36;
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {
      auto fetch = [&](int t, int s) {
        av[s][0] = base[t * 64]; av[s][1] = base[t * 64 + 32];
        for (int n = 0; n < 4; ++n) bv[s][n] = base[4096 + t * 8 + n * 34];
      };
      fetch(0, 0);
#pragma unroll
      for (int t = 0; t < TAPS; ++t) {
        if (t + 1 < TAPS) fetch(t + 1, (t + 1) & 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int n = 0; n < 4; ++n)
            acc[m * 4 + n] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t & 1][m], bv[t & 1][n], acc[m * 4 + n], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else if (MODE == 2 || MODE == 3) {
      // MODE 2: the same reads as three ds_read2_b32 (what the compiler makes of them); MODE 3: additionally a
      // VALU add producing the address of the A reads every tap (offsets beyond the 8-bit range)
      float2 a2[2], b01[2], b23[2];
#define RD2(dst, addr, o0, o1) asm volatile("ds_read2_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(dst) : "v"(addr), "n"(o0), "n"(o1))
      unsigned abase = lbase;
#define FETCH2(t, s)                                                                 \
  if (MODE == 3) asm volatile("v_add_u32 %0, %1, %2" : "=v"(abase) : "v"(lbase), "v"((unsigned)(((t) & 3) * 256)));  \
  RD2(a2[s], abase, 0, 32); RD2(b01[s], lbase, 64 + ((t) & 15), 98 + ((t) & 15)); RD2(b23[s], lbase, 132 + ((t) & 15), 166 + ((t) & 15))
      FETCH2(0, 0);
#pragma unroll
      for (int t = 0; t < TAPS; ++t) {
        if (t + 1 < TAPS) {
          if ((t + 1) & 1) { FETCH2(t + 1, 1); } else { FETCH2(t + 1, 0); }
          asm volatile("s_waitcnt lgkmcnt(3)" ::: "memory");
        } else {
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_sched_barrier(0);
        const float am[2] = {a2[t & 1].x, a2[t & 1].y};
        const float bn[4] = {b01[t & 1].x, b01[t & 1].y, b23[t & 1].x, b23[t & 1].y};
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int n = 0; n < 4; ++n)
            acc[m * 4 + n] = __builtin_amdgcn_mfma_f32_32x32x2f32(am[m], bn[n], acc[m * 4 + n], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
#define RD(dst, off) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(dst) : "v"(lbase), "n"(off))
#define FETCH(t, s)                                                        \
  RD(av[s][0], (t) * 256); RD(av[s][1], (t) * 256 + 128);                  \
  RD(bv[s][0], 16384 + (t) * 32); RD(bv[s][1], 16384 + (t) * 32 + 136);    \
  RD(bv[s][2], 16384 + (t) * 32 + 272); RD(bv[s][3], 16384 + (t) * 32 + 408)
      FETCH(0, 0);
#pragma unroll
      for (int t = 0; t < TAPS; ++t) {
        if (t + 1 < TAPS) {
          if ((t + 1) & 1) { FETCH(t + 1, 1); } else { FETCH(t + 1, 0); }
          asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory");   // the six reads just issued may stay in flight
        } else {
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int n = 0; n < 4; ++n)
            acc[m * 4 + n] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t & 1][m], bv[t & 1][n], acc[m * 4 + n], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
  for (int a = 0; a < 8; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) { clk[blockIdx.x * 2] = t1 - t0; clk[blockIdx.x * 2 + 1] = r1 - r0; }
}

template <int M>
void run(const char* name, int blocks, int iters) {
  float* out; unsigned long long* clk;
  (void)hipMalloc(&out, blocks * 256 * 4); (void)hipMalloc(&clk, blocks * 16);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(k<M>, dim3(blocks), dim3(256), 0, 0, out, clk, iters);
  (void)hipEventRecord(e0, 0);
  const int reps = 10;
  for (int w = 0; w < reps; ++w) hipLaunchKernelGGL(k<M>, dim3(blocks), dim3(256), 0, 0, out, clk, iters);
  (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= reps;
  std::vector<unsigned long long> h(blocks * 2);
  (void)hipMemcpy(h.data(), clk, blocks * 16, hipMemcpyDeviceToHost);
  double cyc = 0, rt = 0; for (int b = 0; b < blocks; ++b) { cyc += h[2 * b]; rt += h[2 * b + 1]; }
  const double flops = (double)blocks * 4 * iters * 36 * 8 * 2.0 * 32 * 32 * 2;
  printf("%-44s blocks %4d: %8.1f us  %6.1f TFLOP/s  clock %.3f GHz  cycles/MFMA/wave %.1f\n", name, blocks, ms * 1e3,
         flops / ms / 1e9, cyc / rt * 0.1, (cyc / blocks) / (iters * 36 * 8.0));
}

int main() {
  const int iters = 120;
  run<0>("compiler-scheduled pipeline, 1 wave/SIMD", 256, iters);
  run<0>("compiler-scheduled pipeline, 2 waves/SIMD", 512, iters);
  run<1>("asm reads + counted waits, 1 wave/SIMD", 256, iters);
  run<1>("asm reads + counted waits, 2 waves/SIMD", 512, iters);
  run<2>("asm ds_read2 + counted waits, 1 wave/SIMD", 256, iters);
  run<3>("asm ds_read2 + VALU address, 1 wave/SIMD", 256, iters);
  return 0;
}
