// HBM-bound geometry / resampling kernels of the DMC path: backward flow warp
// (bilinear gather), 2x2 mean / max pooling, bilinear x2 upsampling, the fused
// SpyNet level input, the squeeze-excite gate and strided copy / clamp.
// One lane per output pixel (consecutive lanes = consecutive x, so loads and
// stores of a wave are contiguous 256-byte rows); channels are looped inside the
// lane so the per-pixel coordinate math is done once for all channels.
#include <atomic>
#include <type_traits>

#include "dvc_common.h"

namespace {

using dvc::stream_grid;

struct Src { const float* p; long long bs, cs; int c; };
struct Dst { float* p; long long bs, cs; };

inline Src mk(const DvcSrc* s) { return Src{s->ptr, s->batch_stride, s->chan_stride, s->channels}; }
inline Dst mk(const DvcDst* d) { return Dst{d->ptr, d->batch_stride, d->chan_stride}; }

// torch.linspace(-1, 1, steps)[idx] in fp32 (symmetric two-sided formula).
__device__ __forceinline__ float linspace_pm1(int idx, int steps) {
  const float step = 2.0f / (float)(steps - 1);
  const int halfway = steps / 2;
  return idx < halfway ? -1.0f + step * (float)idx : 1.0f - step * (float)(steps - idx - 1);
}

struct Bilin {
  int o_nw, o_ne, o_sw, o_se;  // plane offsets (clamped in-bounds)
  int ix, ix1, iy, iy1;        // the same four positions as coordinates
  float nw, ne, sw, se;
};

// Sampling position for output pixel (y,x) displaced by (u,v) pixels, following the
// reference's arithmetic: normalised base grid + flow / ((size-1)/2), then
// grid_sample's un-normalise ((g+1) * (size-1)/2), border clip, floor, weights.
__device__ __forceinline__ Bilin bilin_setup(int y, int x, float u, float v, int h, int w) {
  const float sx = (float)((w - 1.0) / 2.0), sy = (float)((h - 1.0) / 2.0);
  float gx = linspace_pm1(x, w) + u / sx;
  float gy = linspace_pm1(y, h) + v / sy;
  if (w == 1) gx = -1.0f + u / sx;  // linspace with one step is `start`
  if (h == 1) gy = -1.0f + v / sy;
  float fx = (gx + 1.0f) * sx, fy = (gy + 1.0f) * sy;
  fx = fminf((float)(w - 1), fmaxf(fx, 0.0f));
  fy = fminf((float)(h - 1), fmaxf(fy, 0.0f));
  const float xw = floorf(fx), yn = floorf(fy);
  const float we = fx - xw, ee = 1.0f - we, ns = fy - yn, ss = 1.0f - ns;
  Bilin b;
  b.nw = ss * ee; b.ne = ss * we; b.sw = ns * ee; b.se = ns * we;
  const int ix = (int)xw, iy = (int)yn;
  const int ix1 = ix + 1 < w ? ix + 1 : ix;   // out-of-range neighbour has weight 0
  const int iy1 = iy + 1 < h ? iy + 1 : iy;
  if (ix + 1 >= w) { b.ne = 0.f; b.se = 0.f; }
  if (iy + 1 >= h) { b.sw = 0.f; b.se = 0.f; }
  b.o_nw = iy * w + ix; b.o_ne = iy * w + ix1; b.o_sw = iy1 * w + ix; b.o_se = iy1 * w + ix1;
  b.ix = ix; b.ix1 = ix1; b.iy = iy; b.iy1 = iy1;
  return b;
}

// one expression for every kernel that interpolates, so that all of them round alike
// (four products and three sums, each rounded: whether a product is fused into a sum must not depend on what
// the optimiser finds around the call)
__device__ __forceinline__ float bilin_mix(float vnw, float vne, float vsw, float vse, const Bilin& b) {
#pragma clang fp contract(off)
  return vnw * b.nw + vne * b.ne + vsw * b.sw + vse * b.se;
}

__device__ __forceinline__ float bilin_fetch(const float* plane, const Bilin& b) {
  return bilin_mix(plane[b.o_nw], plane[b.o_ne], plane[b.o_sw], plane[b.o_se], b);
}

// The motion-estimation level input interpolates 3-channel images with the products fused into the sums where the
// compiler does so (its results feed only that level's own convolutions; no other kernel has to reproduce them).
__device__ __forceinline__ float bilin_fetch_fused(const float* plane, const Bilin& b) {
  return plane[b.o_nw] * b.nw + plane[b.o_ne] * b.ne + plane[b.o_sw] * b.sw + plane[b.o_se] * b.se;
}

// (Measured and not kept: unrolling the channel loop, or a strip of rows per wave, leaves the time unchanged --
// the gather is bound by vector-cache accesses, about one per cycle and CU, not by latency.  Fetching the west /
// east pair as one 8-byte load at 4-byte alignment was 1.28x faster alone, but under concurrent kernels from other
// streams such misaligned 8-byte loads returned wrong data for whole 16-lane groups in a few percent of launches:
// every multi-dword access in this library is naturally aligned.)
__global__ __launch_bounds__(256) void flow_warp_kernel(Src src, Src flow, Dst out, int batch, int h, int w) {
  const long long hw = (long long)h * w, total = hw * batch;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int n = (int)(i / hw);
    const int pix = (int)(i - n * hw);
    const int y = pix / w, x = pix - y * w;
    const float* fp = flow.p + n * flow.bs + pix;
    const Bilin b = bilin_setup(y, x, fp[0], fp[flow.cs], h, w);
    const float* sp = src.p + n * src.bs;
    float* op = out.p + n * out.bs + pix;
    for (int c = 0; c < src.c; ++c) op[c * out.cs] = bilin_fetch(sp + c * src.cs, b);
  }
}

// Many-channel warp through an LDS window.  The direct gather above is bound by vector-cache accesses (four
// 4-byte gathers per output value), not by bytes.  Here a workgroup owns a 64 x (4 RPL) pixel tile, finds the
// bounding box of the source positions its pixels sample (the flow is shared by all channels), and per channel
// stages that window with aligned 16-byte row loads into one of two LDS buffers while the previous channel is
// interpolated out of the other: one barrier per channel, four LDS reads and one coalesced store per output value.
// A tile whose window does not fit (large or divergent motion) gathers directly, with the same arithmetic.
// Needs w % 4 == 0 and 16-byte aligned planes; blockIdx.y = sample.
constexpr int WARP_WIN_FLOATS = 2560;   // per LDS buffer (10 KiB)

// Workgroups go to the 8 XCDs round-robin in launch order and every XCD has its own L2: renumber the tiles so
// that each XCD works through one contiguous row-major range (neighbouring tiles' windows overlap).
// bx of nt workgroups in a grid row whose first workgroup has launch phase s (= launch index mod 8).
__device__ __forceinline__ int xcd_contiguous_tile(int bx, int nt, int s) {
  const int u = bx + s, k = u & 7;
  const int m1 = s + nt;
  const int before = (m1 >> 3) * k + min(m1 & 7, k) - min(s, k);   // slots of this row that land on XCDs < k
  const int first_u = s + ((k - s) & 7);
  return before + ((u - first_u) >> 3);
}

template <int RPL>
__global__ __launch_bounds__(256) void flow_warp_lds_kernel(Src src, Src flow, Dst out, int h, int w, int tiles_x) {
  constexpr int TH = 4 * RPL;
  constexpr int SPT = (WARP_WIN_FLOATS / 4 + 255) / 256;   // 16-byte staging slots per thread
  __shared__ __attribute__((aligned(16))) float win[2][WARP_WIN_FLOATS];
  __shared__ int red[4][4];
  const int n = blockIdx.y;
  const int tile = xcd_contiguous_tile(blockIdx.x, gridDim.x, (int)((blockIdx.y * gridDim.x) & 7));
  const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int x = tx * 64 + lane, yb = ty * TH + wave * RPL;
  const float* const sp = src.p + n * src.bs;
  const float* const fpn = flow.p + n * flow.bs;
  float* const opn = out.p + n * out.bs;

  Bilin b[RPL];
  bool ok[RPL];
  int minx = 0x7fffffff, maxx = -1, miny = 0x7fffffff, maxy = -1;
#pragma unroll
  for (int r = 0; r < RPL; ++r) {
    const int y = yb + r;
    ok[r] = x < w && y < h;
    const int pix = ok[r] ? y * w + x : 0;
    b[r] = bilin_setup(ok[r] ? y : 0, ok[r] ? x : 0, fpn[pix], fpn[flow.cs + pix], h, w);
    if (ok[r]) {
      minx = min(minx, b[r].ix); maxx = max(maxx, b[r].ix1);
      miny = min(miny, b[r].iy); maxy = max(maxy, b[r].iy1);
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    minx = min(minx, __shfl_xor(minx, off, 64)); maxx = max(maxx, __shfl_xor(maxx, off, 64));
    miny = min(miny, __shfl_xor(miny, off, 64)); maxy = max(maxy, __shfl_xor(maxy, off, 64));
  }
  if (lane == 0) { red[wave][0] = minx; red[wave][1] = maxx; red[wave][2] = miny; red[wave][3] = maxy; }
  __syncthreads();
  minx = min(min(red[0][0], red[1][0]), min(red[2][0], red[3][0]));
  maxx = max(max(red[0][1], red[1][1]), max(red[2][1], red[3][1]));
  miny = min(min(red[0][2], red[1][2]), min(red[2][2], red[3][2]));
  maxy = max(max(red[0][3], red[1][3]), max(red[2][3], red[3][3]));
  if (maxx < 0) return;                      // (no pixel of this tile lies inside the image: never launched)
  const int x0 = minx & ~3, y0 = miny;
  const int wp = (maxx - x0 + 4) & ~3;       // window row pitch: whole 16-byte pieces
  const int hh = maxy - y0 + 1;
  if (wp * hh > WARP_WIN_FLOATS) {           // workgroup-uniform: the window does not fit, gather directly
    for (int c = 0; c < src.c; ++c) {
      const float* const plane = sp + c * src.cs;
      float* const op = opn + c * out.cs;
#pragma unroll
      for (int r = 0; r < RPL; ++r)
        if (ok[r]) op[(yb + r) * w + x] = bilin_fetch(plane, b[r]);
    }
    return;
  }
  // LDS offsets of the four neighbours, staging slots of this thread
  int l_nw[RPL], l_ne[RPL], l_sw[RPL], l_se[RPL];
#pragma unroll
  for (int r = 0; r < RPL; ++r) {
    const int rn = (b[r].iy - y0) * wp, rs = (b[r].iy1 - y0) * wp;
    l_nw[r] = ok[r] ? rn + b[r].ix - x0 : 0; l_ne[r] = ok[r] ? rn + b[r].ix1 - x0 : 0;
    l_sw[r] = ok[r] ? rs + b[r].ix - x0 : 0; l_se[r] = ok[r] ? rs + b[r].ix1 - x0 : 0;
  }
  const int w4 = wp >> 2, nslots = w4 * hh;
  int goff[SPT];
  bool sok[SPT];
#pragma unroll
  for (int s = 0; s < SPT; ++s) {
    const int idx = tid + 256 * s;
    const int row = idx / w4, col = idx - row * w4;
    const int gx = x0 + 4 * col;
    sok[s] = idx < nslots && gx < w;       // (w % 4 == 0: a piece is wholly inside a row or wholly past its end)
    goff[s] = sok[s] ? (y0 + row) * w + gx : 0;
  }
  static_assert(SPT == 3, "the staging registers below are written out for three slots per thread");
  float4 st0, st1, st2;   // (named registers: an array here ends up in private memory)
#define DVC_WARP_FETCH(c_)                                                   \
  do {                                                                       \
    const float* const plane_ = sp + (c_) * src.cs;                          \
    st0 = *reinterpret_cast<const float4*>(plane_ + goff[0]);                \
    st1 = *reinterpret_cast<const float4*>(plane_ + goff[1]);                \
    st2 = *reinterpret_cast<const float4*>(plane_ + goff[2]);                \
  } while (0)   /* (slots without a piece re-read the plane's first 16 bytes) */
#define DVC_WARP_COMMIT(buf_)                                                \
  do {                                                                       \
    float4* const dst_ = reinterpret_cast<float4*>(win[buf_]) + tid;        \
    if (sok[0]) dst_[0] = st0;                                               \
    if (sok[1]) dst_[256] = st1;                                             \
    if (sok[2]) dst_[512] = st2;                                             \
  } while (0)
  // (Requesting a window two channels ahead, or barriers that wait for LDS only, change nothing: with loads and
  // stores of one wave in flight together the hardware's single vector-memory counter cannot tell them apart, so
  // every wait for a window load also drains the wave's stores.)
  DVC_WARP_FETCH(0);
  DVC_WARP_COMMIT(0);
  __syncthreads();
  for (int c = 0; c < src.c; ++c) {
    const bool more = c + 1 < src.c;
    if (more) DVC_WARP_FETCH(c + 1);
    const float* const wv = win[c & 1];
    float* const op = opn + c * out.cs;
#pragma unroll
    for (int r = 0; r < RPL; ++r) {
      const float v = bilin_mix(wv[l_nw[r]], wv[l_ne[r]], wv[l_sw[r]], wv[l_se[r]], b[r]);
      if (ok[r]) op[(yb + r) * w + x] = v;
    }
    if (more) DVC_WARP_COMMIT((c + 1) & 1);
    __syncthreads();
  }
#undef DVC_WARP_FETCH
#undef DVC_WARP_COMMIT
}

// The same warp with the two halves of the work on different waves.  In the kernel above one wave both requests
// window rows and stores output rows, and the hardware counts a wave's loads and stores in ONE vector-memory
// counter: every wait for a window also waits for the stores issued before it.  Here waves 4..7 only bring windows
// into a ring of three LDS buffers (global -> LDS DMA, 16 bytes per lane, no register round trip; the window of
// channel c + 2 is requested while channel c is interpolated) and wait for nothing but their own requests; waves
// 0..3 only read LDS and store (per-lane addresses), and never wait for memory.  One bare barrier per channel, taken by all eight waves.
// Same tile, same window, same arithmetic as above: bit-identical output.  Needs what the kernel above needs, and
// channel offsets that fit the buffer instructions' 32-bit scalar offset.
constexpr int WARP_RING = 3;

// One LDS dword at (per-lane byte address + immediate).  Issues the read and returns: the value is valid only
// behind an s_waitcnt lgkmcnt placed by the caller.
template <int OFFSET_BYTES>
__device__ __forceinline__ float warp_lds_read(unsigned addr) {
  static_assert(OFFSET_BYTES >= 0 && OFFSET_BYTES < 65536, "DS instructions carry a 16-bit offset");
  float v;
  asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFFSET_BYTES));
  return v;
}

template <int RPL>
__global__ __launch_bounds__(512) void flow_warp_split_kernel(Src src, Src flow, Dst out, int h, int w, int tiles_x) {
  constexpr int TH = 4 * RPL;
  constexpr int SPT = (WARP_WIN_FLOATS / 4 + 255) / 256;   // 16-byte staging slots per loader lane
  // (a buffer holds whole rounds of 256 pieces: slots past the window receive zeros instead of being skipped)
  constexpr int BUF_FLOATS = SPT * 1024;
  __shared__ __attribute__((aligned(16))) float win[WARP_RING][BUF_FLOATS];
  __shared__ int red[4][4];
  const int n = blockIdx.y;
  const int tile = xcd_contiguous_tile(blockIdx.x, gridDim.x, (int)((blockIdx.y * gridDim.x) & 7));
  const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
  const int tid = threadIdx.x & 255, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool loader = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 8)) != 0;   // wave-uniform
  const int x = tx * 64 + lane, yb = ty * TH + wave * RPL;
  const float* const sp = src.p + n * src.bs;
  const float* const fpn = flow.p + n * flow.bs;
  float* const opn = out.p + n * out.bs;

  Bilin b[RPL];
  bool ok[RPL];
  if (!loader) {
    int minx = 0x7fffffff, maxx = -1, miny = 0x7fffffff, maxy = -1;
#pragma unroll
    for (int r = 0; r < RPL; ++r) {
      const int y = yb + r;
      ok[r] = x < w && y < h;
      const int pix = ok[r] ? y * w + x : 0;
      b[r] = bilin_setup(ok[r] ? y : 0, ok[r] ? x : 0, fpn[pix], fpn[flow.cs + pix], h, w);
      if (ok[r]) {
        minx = min(minx, b[r].ix); maxx = max(maxx, b[r].ix1);
        miny = min(miny, b[r].iy); maxy = max(maxy, b[r].iy1);
      }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      minx = min(minx, __shfl_xor(minx, off, 64)); maxx = max(maxx, __shfl_xor(maxx, off, 64));
      miny = min(miny, __shfl_xor(miny, off, 64)); maxy = max(maxy, __shfl_xor(maxy, off, 64));
    }
    if (lane == 0) { red[wave][0] = minx; red[wave][1] = maxx; red[wave][2] = miny; red[wave][3] = maxy; }
  }
  __syncthreads();
  const int minx = min(min(red[0][0], red[1][0]), min(red[2][0], red[3][0]));
  const int maxx = max(max(red[0][1], red[1][1]), max(red[2][1], red[3][1]));
  const int miny = min(min(red[0][2], red[1][2]), min(red[2][2], red[3][2]));
  const int maxy = max(max(red[0][3], red[1][3]), max(red[2][3], red[3][3]));
  if (maxx < 0) return;
  const int x0 = minx & ~3, y0 = miny;
  const int wp = (maxx - x0 + 4) & ~3;
  const int hh = maxy - y0 + 1;
  const int nc = src.c;
  if (wp * hh > WARP_WIN_FLOATS) {           // workgroup-uniform: the window does not fit, gather directly
    if (loader) return;
    for (int c = 0; c < nc; ++c) {
      const float* const plane = sp + c * src.cs;
      float* const op = opn + c * out.cs;
#pragma unroll
      for (int r = 0; r < RPL; ++r)
        if (ok[r]) op[(yb + r) * w + x] = bilin_fetch(plane, b[r]);
    }
    return;
  }
  // both roles pass exactly nc + 1 barriers from here on
  if (loader) {
    const int w4 = wp >> 2, nslots = w4 * hh;
    unsigned voff[SPT];
#pragma unroll
    for (int s = 0; s < SPT; ++s) {
      const int idx = tid + 256 * s;
      const int row = idx / w4, col = idx - row * w4;
      const int gx = x0 + 4 * col;
      const bool sok = idx < nslots && gx < w;
      voff[s] = sok ? 4u * (unsigned)((y0 + row) * w + gx) : 0x80000000u;   // past the range: the slot receives zeros
    }
    const unsigned cs4 = __builtin_amdgcn_readfirstlane(4u * (unsigned)src.cs);
    const unsigned plane_bytes = 4u * (unsigned)h * (unsigned)w;
    auto request = [&](int c) __attribute__((always_inline)) {
      float* const dst = win[c % WARP_RING] + 256 * wave;   // this wave's 64 pieces of every 256
      // the channel is the scalar offset of the load; the range check compares the per-lane offset with
      // (records - scalar offset), so the records end with this channel's plane
      const unsigned soff = (unsigned)c * cs4;
      const __amdgpu_buffer_rsrc_t rsrc =
          __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(sp), 0, soff + plane_bytes, 0x00020000);
#pragma unroll
      for (int s = 0; s < SPT; ++s)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(dst + 1024 * s), 16,
                                                 voff[s], soff, 0, 0);
    };
    constexpr int kVm0 = 0x0f70, kVmSPT = 0x0f70 | SPT;   // vmcnt(0) / vmcnt(SPT), nothing else waited for
    request(0);
    if (nc > 1) request(1);
    if (nc > 1) __builtin_amdgcn_s_waitcnt(kVmSPT); else __builtin_amdgcn_s_waitcnt(kVm0);
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    for (int c = 0; c < nc; ++c) {
      // buffer (c + 2) % 3 held channel c - 1: the interpolating waves left it before the barrier just passed
      if (c + 2 < nc) {
        request(c + 2);
        __builtin_amdgcn_s_waitcnt(kVmSPT);    // channel c + 1 has landed
      } else {
        __builtin_amdgcn_s_waitcnt(kVm0);
      }
      asm volatile("" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
    // (ends the wave here, so that no path leads from the DMA requests to the other role's LDS reads: the
    // compiler would otherwise make every one of those reads wait for all vector-memory traffic of its wave)
    __builtin_amdgcn_endpgm();
  }
  int l_nw[RPL], l_ne[RPL], l_sw[RPL], l_se[RPL];
#pragma unroll
  for (int r = 0; r < RPL; ++r) {
    const int rn = (b[r].iy - y0) * wp, rs = (b[r].iy1 - y0) * wp;
    l_nw[r] = ok[r] ? rn + b[r].ix - x0 : 0; l_ne[r] = ok[r] ? rn + b[r].ix1 - x0 : 0;
    l_sw[r] = ok[r] ? rs + b[r].ix - x0 : 0; l_se[r] = ok[r] ? rs + b[r].ix1 - x0 : 0;
  }
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  // LDS byte addresses of the four neighbours in ring buffer 0; the buffer of a channel is an immediate offset.
  // The reads are explicit instructions the compiler does not see as LDS accesses: for a read it does see, it
  // first waits until every store the wave has issued is complete (loads and stores share one counter here).
  const unsigned win0 = (unsigned)(size_t)(__attribute__((address_space(3))) const float*)&win[0][0];
  unsigned a_nw[RPL], a_ne[RPL], a_sw[RPL], a_se[RPL];
#pragma unroll
  for (int r = 0; r < RPL; ++r) {
    a_nw[r] = win0 + 4u * (unsigned)l_nw[r]; a_ne[r] = win0 + 4u * (unsigned)l_ne[r];
    a_sw[r] = win0 + 4u * (unsigned)l_sw[r]; a_se[r] = win0 + 4u * (unsigned)l_se[r];
  }
  static_assert(RPL == 4, "the wait below names the sixteen values of four rows");
  auto channel = [&](int c, auto slot) __attribute__((always_inline)) {
    constexpr int OFF = decltype(slot)::value * BUF_FLOATS * 4;
    float vnw[RPL], vne[RPL], vsw[RPL], vse[RPL];
#pragma unroll
    for (int r = 0; r < RPL; ++r) {
      vnw[r] = warp_lds_read<OFF>(a_nw[r]); vne[r] = warp_lds_read<OFF>(a_ne[r]);
      vsw[r] = warp_lds_read<OFF>(a_sw[r]); vse[r] = warp_lds_read<OFF>(a_se[r]);
    }
    // the values exist behind this wait; both statements name them as in-out so that no use can move above
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(vnw[0]), "+v"(vne[0]), "+v"(vsw[0]), "+v"(vse[0]), "+v"(vnw[1]), "+v"(vne[1]), "+v"(vsw[1]),
                   "+v"(vse[1]));
    asm volatile(""
                 : "+v"(vnw[2]), "+v"(vne[2]), "+v"(vsw[2]), "+v"(vse[2]), "+v"(vnw[3]), "+v"(vne[3]), "+v"(vsw[3]),
                   "+v"(vse[3]));
    // (Plain predicated stores.  A branch-free form -- buffer stores with the channel as scalar offset and a
    // descriptor whose size grows with it, a pixel outside the image dropped by the range check -- was bit-identical
    // alone and wrong in lanes 48..63 of some waves in 40 % of the launches while another stream kept the GPU busy.
    // Cause not established; what the two forms differ in is that the buffer store's scalar operands are rewritten
    // for the next channel a few instructions after it is issued.)
    float* const op = opn + c * out.cs;
#pragma unroll
    for (int r = 0; r < RPL; ++r) {
      const float v = bilin_mix(vnw[r], vne[r], vsw[r], vse[r], b[r]);
      if (ok[r]) op[(yb + r) * w + x] = v;
    }
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };
  static_assert(WARP_RING == 3 && 3 * BUF_FLOATS * 4 < 65536, "ring slots as immediate offsets of the reads");
  for (int c = 0; c < nc; c += 3) {   // (nc is workgroup-uniform: every wave takes the same barriers)
    channel(c, std::integral_constant<int, 0>{});
    if (c + 1 < nc) channel(c + 1, std::integral_constant<int, 1>{});
    if (c + 2 < nc) channel(c + 2, std::integral_constant<int, 2>{});
  }
}

// Flat index over (n, c, oy, ox) of the output.
#define DVC_FOREACH_OUT(total_expr)                                                    \
  const long long total_ = (total_expr);                                                \
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total_; i += (long long)gridDim.x * 256)

__global__ __launch_bounds__(256) void avg_pool2_kernel(Src src, Dst out, int batch, int h, int w, float scale) {
  const int ho = h / 2, wo = w / 2;
  DVC_FOREACH_OUT((long long)batch * src.c * ho * wo) {
    const int ox = (int)(i % wo);
    long long t = i / wo;
    const int oy = (int)(t % ho); t /= ho;
    const int c = (int)(t % src.c), n = (int)(t / src.c);
    const float* p = src.p + n * src.bs + c * src.cs + (long long)(2 * oy) * w + 2 * ox;
    // bilinear half-pixel weights are all .5: .5*(.5a+.5b) + .5*(.5c+.5d)
    const float v = 0.5f * (0.5f * p[0] + 0.5f * p[1]) + 0.5f * (0.5f * p[w] + 0.5f * p[w + 1]);
    out.p[n * out.bs + c * out.cs + (long long)oy * wo + ox] = v * scale;
  }
}

__global__ __launch_bounds__(256) void max_pool2_kernel(Src src, Dst out, int batch, int h, int w) {
  const int ho = h / 2, wo = w / 2;
  DVC_FOREACH_OUT((long long)batch * src.c * ho * wo) {
    const int ox = (int)(i % wo);
    long long t = i / wo;
    const int oy = (int)(t % ho); t /= ho;
    const int c = (int)(t % src.c), n = (int)(t / src.c);
    const float* p = src.p + n * src.bs + c * src.cs + (long long)(2 * oy) * w + 2 * ox;
    out.p[n * out.bs + c * out.cs + (long long)oy * wo + ox] = fmaxf(fmaxf(p[0], p[1]), fmaxf(p[w], p[w + 1]));
  }
}

// align_corners=False x2: src = .5*(dst+.5)-.5 clamped at 0; i1 = min(i0+1, in-1).
__device__ __forceinline__ void up2_coord(int o, int in, int& i0, int& i1, float& l0, float& l1) {
  float s = 0.5f * ((float)o + 0.5f) - 0.5f;
  s = s < 0.f ? 0.f : s;
  i0 = (int)s;
  i1 = i0 + (i0 < in - 1 ? 1 : 0);
  l1 = s - (float)i0;
  l0 = 1.0f - l1;
}

__device__ __forceinline__ float up2_sample(const float* plane, int w_in, int y0, int y1, float ly0, float ly1,
                                            int x0, int x1, float lx0, float lx1) {
  return ly0 * (lx0 * plane[y0 * w_in + x0] + lx1 * plane[y0 * w_in + x1]) +
         ly1 * (lx0 * plane[y1 * w_in + x0] + lx1 * plane[y1 * w_in + x1]);
}

__global__ __launch_bounds__(256) void bilinear_up2_kernel(Src src, Dst out, int batch, int h, int w, float scale) {
  const int ho = 2 * h, wo = 2 * w;
  DVC_FOREACH_OUT((long long)batch * src.c * ho * wo) {
    const int ox = (int)(i % wo);
    long long t = i / wo;
    const int oy = (int)(t % ho); t /= ho;
    const int c = (int)(t % src.c), n = (int)(t / src.c);
    int x0, x1, y0, y1; float lx0, lx1, ly0, ly1;
    up2_coord(ox, w, x0, x1, lx0, lx1);
    up2_coord(oy, h, y0, y1, ly0, ly1);
    const float* p = src.p + n * src.bs + c * src.cs;
    out.p[n * out.bs + c * out.cs + (long long)oy * wo + ox] =
        up2_sample(p, w, y0, y1, ly0, ly1, x0, x1, lx0, lx1) * scale;
  }
}

__global__ __launch_bounds__(256) void spynet_level_input_kernel(Src im1, Src im2, Src flow_prev, Dst out,
                                                                int batch, int h, int w) {
  const long long hw = (long long)h * w, total = hw * batch;
  const int hp = h / 2, wp = w / 2;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int n = (int)(i / hw);
    const int pix = (int)(i - n * hw);
    const int y = pix / w, x = pix - y * w;
    float u = 0.f, v = 0.f;
    if (flow_prev.p != nullptr) {
      int x0, x1, y0, y1; float lx0, lx1, ly0, ly1;
      up2_coord(x, wp, x0, x1, lx0, lx1);
      up2_coord(y, hp, y0, y1, ly0, ly1);
      const float* fp = flow_prev.p + n * flow_prev.bs;
      u = up2_sample(fp, wp, y0, y1, ly0, ly1, x0, x1, lx0, lx1) * 2.0f;
      v = up2_sample(fp + flow_prev.cs, wp, y0, y1, ly0, ly1, x0, x1, lx0, lx1) * 2.0f;
    }
    const Bilin b = bilin_setup(y, x, u, v, h, w);
    float* op = out.p + n * out.bs + pix;
    const float* p1 = im1.p + n * im1.bs + pix;
    const float* p2 = im2.p + n * im2.bs;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      op[c * out.cs] = p1[c * im1.cs];
      op[(3 + c) * out.cs] = bilin_fetch_fused(p2 + c * im2.cs, b);
    }
    op[6 * out.cs] = u;
    op[7 * out.cs] = v;
  }
}

__global__ __launch_bounds__(256) void copy_kernel(Src src, Dst out, int batch, int h, int w) {
  const long long hw = (long long)h * w;
  DVC_FOREACH_OUT((long long)batch * src.c * hw) {
    const long long pix = i % hw;
    long long t = i / hw;
    const int c = (int)(t % src.c), n = (int)(t / src.c);
    out.p[n * out.bs + c * out.cs + pix] = src.p[n * src.bs + c * src.cs + pix];
  }
}

// the same in 16-byte units (planes of a multiple of 4 floats, 16-byte aligned pointers and strides)
__global__ __launch_bounds__(256) void copy4_kernel(Src src, Dst out, unsigned chans, unsigned plane4, unsigned total) {
  for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
    const unsigned p4 = i % plane4, q = i / plane4, c = q % chans, n = q / chans;
    reinterpret_cast<float4*>(out.p + (long long)n * out.bs + (long long)c * out.cs)[p4] =
        reinterpret_cast<const float4*>(src.p + (long long)n * src.bs + (long long)c * src.cs)[p4];
  }
}

__global__ __launch_bounds__(256) void clamp_kernel(Src src, Dst out, int batch, int h, int w, float lo, float hi) {
  const long long hw = (long long)h * w;
  DVC_FOREACH_OUT((long long)batch * src.c * hw) {
    const long long pix = i % hw;
    long long t = i / hw;
    const int c = (int)(t % src.c), n = (int)(t / src.c);
    const float v = src.p[n * src.bs + c * src.cs + pix];
    out.p[n * out.bs + c * out.cs + pix] = fminf(fmaxf(v, lo), hi);
  }
}

// ---------------------------------------------------------------- squeeze-excite
constexpr int SE_SLICES = 64;  // partial sums per (n, c) plane

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

// Sum of 256 per-thread values, result valid in thread 0.  Fixed order.
__device__ __forceinline__ float block_sum_256(float v, float* lds4) {
  v = wave_sum(v);
  if ((threadIdx.x & 63) == 0) lds4[threadIdx.x >> 6] = v;
  __syncthreads();
  float r = 0.f;
  if (threadIdx.x == 0) r = (lds4[0] + lds4[1]) + (lds4[2] + lds4[3]);
  __syncthreads();
  return r;
}

// VEC: plane bases are 16-byte aligned (slices are whole float4s apart): 16-byte loads, four independent
// accumulators per thread.  The summation order is fixed either way.
template <bool VEC>
__global__ __launch_bounds__(256) void plane_partial_sum_kernel(Src x, int h, int w, float* partial) {
  __shared__ float red[4];
  const int slice = blockIdx.x, c = blockIdx.y, n = blockIdx.z;
  const long long hw = (long long)h * w;
  const long long per = ((hw + SE_SLICES - 1) / SE_SLICES + 3) & ~3LL;
  const long long b = slice * per < hw ? slice * per : hw, e = b + per < hw ? b + per : hw;
  const float* p = x.p + n * x.bs + c * x.cs;
  float s = 0.f;
  if constexpr (VEC) {
    const float4* p4 = reinterpret_cast<const float4*>(p + b);
    const int n4 = (int)((e - b) >> 2);
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int i = threadIdx.x;
    for (; i + 768 < n4; i += 1024) {
      const float4 v0 = p4[i], v1 = p4[i + 256], v2 = p4[i + 512], v3 = p4[i + 768];
      a0 += (v0.x + v0.y) + (v0.z + v0.w);
      a1 += (v1.x + v1.y) + (v1.z + v1.w);
      a2 += (v2.x + v2.y) + (v2.z + v2.w);
      a3 += (v3.x + v3.y) + (v3.z + v3.w);
    }
    for (; i < n4; i += 256) {
      const float4 v0 = p4[i];
      a0 += (v0.x + v0.y) + (v0.z + v0.w);
    }
    s = (a0 + a1) + (a2 + a3);
    for (long long j = b + 4LL * n4 + threadIdx.x; j < e; j += 256) s += p[j];   // (last slice only: < 4 elements)
  } else {
    for (long long i = b + threadIdx.x; i < e; i += 256) s += p[i];
  }
  s = block_sum_256(s, red);
  if (threadIdx.x == 0) partial[((long long)n * x.c + c) * SE_SLICES + slice] = s;
}

void launch_plane_partial_sums(const DvcSrc* x, int batch, int h, int w, float* partial, hipStream_t st) {
  const bool vec = reinterpret_cast<uintptr_t>(x->ptr) % 16 == 0 && x->batch_stride % 4 == 0 && x->chan_stride % 4 == 0;
  const dim3 grid(SE_SLICES, x->channels, batch);
  if (vec) hipLaunchKernelGGL(plane_partial_sum_kernel<true>, grid, dim3(256), 0, st, mk(x), h, w, partial);
  else hipLaunchKernelGGL(plane_partial_sum_kernel<false>, grid, dim3(256), 0, st, mk(x), h, w, partial);
}

// Second stage for plane sums that came out of a convolution's epilogue (one value per workgroup tile, `slots` per
// plane): one wave per plane folds them into the SE_SLICES partial sums se_fc_kernel reads -- lane l takes the slots
// congruent to l, so the order is fixed.
__global__ __launch_bounds__(64) void fold_plane_sums_kernel(const float* sums, int slots, float* partial) {
  const long long plane = blockIdx.x;   // n * channels + c
  const float* p = sums + plane * slots;
  float s = 0.f;
  for (int k = threadIdx.x; k < slots; k += SE_SLICES) s += p[k];
  partial[plane * SE_SLICES + threadIdx.x] = s;
}

// one block per n: mean -> fc1 -> relu -> fc2 -> sigmoid
__global__ __launch_bounds__(256) void se_fc_kernel(const float* partial, int channels, int hidden, float inv_hw,
                                                    const float* fc1, const float* fc2, float* gate) {
  __shared__ float mean_s[256];
  __shared__ float hid_s[64];
  const int n = blockIdx.x, t = threadIdx.x;
  if (t < channels) {
    const float* pp = partial + ((long long)n * channels + t) * SE_SLICES;
    float s = 0.f;
    for (int k = 0; k < SE_SLICES; ++k) s += pp[k];
    mean_s[t] = s * inv_hw;
  }
  __syncthreads();
  if (t < hidden) {
    float s = 0.f;
    for (int c = 0; c < channels; ++c) s += fc1[t * channels + c] * mean_s[c];
    hid_s[t] = s > 0.f ? s : 0.f;
  }
  __syncthreads();
  if (t < channels) {
    float s = 0.f;
    for (int j = 0; j < hidden; ++j) s += fc2[t * hidden + j] * hid_s[j];
    gate[(long long)n * channels + t] = 1.0f / (1.0f + expf(-s));
  }
}

int check_plane(const DvcSrc* s, int h, int w, const char* who) {
  DVC_REQUIRE(s != nullptr && s->ptr != nullptr && s->channels >= 1, "%s: null/empty source", who);
  DVC_REQUIRE(s->chan_stride >= (int64_t)h * w, "%s: channel stride < H*W", who);
  return DVC_OK;
}

}  // namespace

#define DVC_CHECK_SRC(s, h, w, who) do { int rc_ = check_plane((s), (h), (w), (who)); if (rc_) return rc_; } while (0)
#define DVC_CHECK_DST(d, plane, who)                                                   \
  DVC_REQUIRE((d) != nullptr && (d)->ptr != nullptr && (d)->chan_stride >= (int64_t)(plane), "%s: bad destination", who)

extern "C" int dvc_flow_warp(const DvcSrc* src, const DvcSrc* flow, const DvcDst* out, int32_t batch, int32_t h,
                             int32_t w, void* stream) {
  DVC_REQUIRE(batch >= 1 && h >= 1 && w >= 1, "dvc_flow_warp: bad dims");
  DVC_CHECK_SRC(src, h, w, "dvc_flow_warp(src)");
  DVC_CHECK_SRC(flow, h, w, "dvc_flow_warp(flow)");
  DVC_REQUIRE(flow->channels == 2, "dvc_flow_warp: flow must have 2 channels");
  DVC_CHECK_DST(out, (int64_t)h * w, "dvc_flow_warp");
  DVC_REQUIRE((int64_t)h * w < (1LL << 30), "dvc_flow_warp: plane too large");
  auto al16 = [](const void* p, long long bs, long long cs) {
    return reinterpret_cast<uintptr_t>(p) % 16 == 0 && bs % 4 == 0 && cs % 4 == 0;
  };
  if (src->channels >= 16 && w % 4 == 0 && batch <= 65535 && al16(src->ptr, src->batch_stride, src->chan_stride)) {
    // many channels on a plane large enough to give every CU a 64 x 16 tile: the LDS-window kernel (64 channels
    // at 1088x1920: 460 -> 268 us, 4.0 TB/s of algorithmic bytes; at 272x480 the direct gather is the faster one)
    const int tiles_x = (w + 63) / 64;
    const long long tiles = (long long)tiles_x * ((h + 15) / 16);
    if (tiles * batch >= 256 && tiles < (1LL << 31) && ((long long)src->channels + 1) * src->chan_stride * 4 < (1LL << 31) &&
        ((long long)src->channels + 1) * out->chan_stride * 4 < (1LL << 31)) {
      hipLaunchKernelGGL(flow_warp_split_kernel<4>, dim3((unsigned)tiles, batch), dim3(512), 0, (hipStream_t)stream,
                         mk(src), mk(flow), mk(out), h, w, tiles_x);
      return dvc::check_launch("flow_warp_split_kernel");
    }
    if (tiles * batch >= 256 && tiles < (1LL << 31)) {
      hipLaunchKernelGGL(flow_warp_lds_kernel<4>, dim3((unsigned)tiles, batch), dim3(256), 0, (hipStream_t)stream,
                         mk(src), mk(flow), mk(out), h, w, tiles_x);
      return dvc::check_launch("flow_warp_lds_kernel");
    }
  }
  const long long total = (long long)batch * h * w;
  hipLaunchKernelGGL(flow_warp_kernel, dim3(stream_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, mk(src),
                     mk(flow), mk(out), batch, h, w);
  return dvc::check_launch("flow_warp_kernel");
}

extern "C" int dvc_avg_pool2(const DvcSrc* src, const DvcDst* out, int32_t batch, int32_t h, int32_t w, float scale,
                             void* stream) {
  DVC_REQUIRE(batch >= 1 && h >= 2 && w >= 2 && h % 2 == 0 && w % 2 == 0, "dvc_avg_pool2: H, W must be even");
  DVC_CHECK_SRC(src, h, w, "dvc_avg_pool2");
  DVC_CHECK_DST(out, (int64_t)(h / 2) * (w / 2), "dvc_avg_pool2");
  const long long total = (long long)batch * src->channels * (h / 2) * (w / 2);
  hipLaunchKernelGGL(avg_pool2_kernel, dim3(stream_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, mk(src),
                     mk(out), batch, h, w, scale);
  return dvc::check_launch("avg_pool2_kernel");
}

extern "C" int dvc_max_pool2(const DvcSrc* src, const DvcDst* out, int32_t batch, int32_t h, int32_t w,
                             void* stream) {
  DVC_REQUIRE(batch >= 1 && h >= 2 && w >= 2 && h % 2 == 0 && w % 2 == 0, "dvc_max_pool2: H, W must be even");
  DVC_CHECK_SRC(src, h, w, "dvc_max_pool2");
  DVC_CHECK_DST(out, (int64_t)(h / 2) * (w / 2), "dvc_max_pool2");
  const long long total = (long long)batch * src->channels * (h / 2) * (w / 2);
  hipLaunchKernelGGL(max_pool2_kernel, dim3(stream_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, mk(src),
                     mk(out), batch, h, w);
  return dvc::check_launch("max_pool2_kernel");
}

extern "C" int dvc_bilinear_up2(const DvcSrc* src, const DvcDst* out, int32_t batch, int32_t h, int32_t w,
                                float scale, void* stream) {
  DVC_REQUIRE(batch >= 1 && h >= 1 && w >= 1, "dvc_bilinear_up2: bad dims");
  DVC_CHECK_SRC(src, h, w, "dvc_bilinear_up2");
  DVC_CHECK_DST(out, (int64_t)4 * h * w, "dvc_bilinear_up2");
  const long long total = (long long)batch * src->channels * 4 * h * w;
  hipLaunchKernelGGL(bilinear_up2_kernel, dim3(stream_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, mk(src),
                     mk(out), batch, h, w, scale);
  return dvc::check_launch("bilinear_up2_kernel");
}

extern "C" int dvc_spynet_level_input(const DvcSrc* im1, const DvcSrc* im2, const DvcSrc* flow_prev,
                                      const DvcDst* out, int32_t batch, int32_t h, int32_t w, void* stream) {
  DVC_REQUIRE(batch >= 1 && h >= 2 && w >= 2 && h % 2 == 0 && w % 2 == 0, "dvc_spynet_level_input: H, W must be even");
  DVC_CHECK_SRC(im1, h, w, "dvc_spynet_level_input(im1)");
  DVC_CHECK_SRC(im2, h, w, "dvc_spynet_level_input(im2)");
  DVC_REQUIRE(im1->channels == 3 && im2->channels == 3, "dvc_spynet_level_input: images must have 3 channels");
  Src fp{nullptr, 0, 0, 2};
  if (flow_prev != nullptr && flow_prev->ptr != nullptr) {
    DVC_CHECK_SRC(flow_prev, h / 2, w / 2, "dvc_spynet_level_input(flow_prev)");
    DVC_REQUIRE(flow_prev->channels == 2, "dvc_spynet_level_input: flow must have 2 channels");
    fp = mk(flow_prev);
  }
  DVC_CHECK_DST(out, (int64_t)h * w, "dvc_spynet_level_input");
  const long long total = (long long)batch * h * w;
  hipLaunchKernelGGL(spynet_level_input_kernel, dim3(stream_grid(total, 256)), dim3(256), 0, (hipStream_t)stream,
                     mk(im1), mk(im2), fp, mk(out), batch, h, w);
  return dvc::check_launch("spynet_level_input_kernel");
}

extern "C" int64_t dvc_se_scratch_floats(int32_t batch, int32_t channels) {
  return (int64_t)batch * channels * SE_SLICES;
}

extern "C" int dvc_se_gate(const DvcSrc* x, int32_t batch, int32_t h, int32_t w, const float* fc1, const float* fc2,
                           int32_t hidden, float* partial, float* gate, void* stream) {
  DVC_REQUIRE(batch >= 1 && h >= 1 && w >= 1, "dvc_se_gate: bad dims");
  DVC_CHECK_SRC(x, h, w, "dvc_se_gate");
  DVC_REQUIRE(x->channels <= 256 && hidden >= 1 && hidden <= 64, "dvc_se_gate: channels<=256, hidden<=64 required");
  DVC_REQUIRE(fc1 && fc2 && partial && gate, "dvc_se_gate: null pointer");
  launch_plane_partial_sums(x, batch, h, w, partial, (hipStream_t)stream);
  int rc = dvc::check_launch("plane_partial_sum_kernel");
  if (rc) return rc;
  hipLaunchKernelGGL(se_fc_kernel, dim3(batch), dim3(256), 0, (hipStream_t)stream, (const float*)partial, x->channels,
                     hidden, 1.0f / ((float)h * (float)w), fc1, fc2, gate);
  return dvc::check_launch("se_fc_kernel");
}

extern "C" int dvc_se_gate_from_sums(const float* sums, int32_t slots, int32_t batch, int32_t channels, int32_t h,
                                     int32_t w, const float* fc1, const float* fc2, int32_t hidden, float* partial,
                                     float* gate, void* stream) {
  static_assert(SE_SLICES == 64, "fold_plane_sums_kernel: one lane per slice");
  DVC_REQUIRE(batch >= 1 && h >= 1 && w >= 1 && slots >= 1, "dvc_se_gate_from_sums: bad dims");
  DVC_REQUIRE(channels >= 1 && channels <= 256 && hidden >= 1 && hidden <= 64,
              "dvc_se_gate_from_sums: channels<=256, hidden<=64 required");
  DVC_REQUIRE(sums && fc1 && fc2 && partial && gate, "dvc_se_gate_from_sums: null pointer");
  hipLaunchKernelGGL(fold_plane_sums_kernel, dim3(batch * channels), dim3(64), 0, (hipStream_t)stream, sums, slots,
                     partial);
  int rc = dvc::check_launch("fold_plane_sums_kernel");
  if (rc) return rc;
  hipLaunchKernelGGL(se_fc_kernel, dim3(batch), dim3(256), 0, (hipStream_t)stream, (const float*)partial, channels,
                     hidden, 1.0f / ((float)h * (float)w), fc1, fc2, gate);
  return dvc::check_launch("se_fc_kernel");
}

// ---- squeeze-excite gate without any allocation: the partial sums live in a file-scope device array.
// Each GPU the library runs on holds its own copy of the array (at its own address) until the process ends.
constexpr int SE_STATIC_MAX_PLANES = DVC_SE_STATIC_MAX_PLANES;
static __device__ __align__(256) float scratch[SE_STATIC_MAX_PLANES * SE_SLICES];   // 1 MiB

namespace {
// Device address of `scratch` per GPU.  In host code the array's name is only a handle for the symbol
// lookup, not a device address: kernels get the looked-up pointer as an argument.
std::atomic<float*> g_scratch_ptr[DVC_MAX_DEVICES];

int static_scratch(float** out) {
  int device = -1;
  hipError_t e = hipGetDevice(&device);
  if (e != hipSuccess || device < 0 || device >= DVC_MAX_DEVICES) {
    dvc::set_error("dvc_se_gate_static: no current device (%s)", e != hipSuccess ? hipGetErrorString(e) : "index out of range");
    return DVC_EHIP;
  }
  float* p = g_scratch_ptr[device].load(std::memory_order_acquire);
  if (p == nullptr) {   // first call on this device (two threads racing here look up the same address)
    void* sym = nullptr;
    e = hipGetSymbolAddress(&sym, HIP_SYMBOL(scratch));
    if (e != hipSuccess || sym == nullptr) {
      dvc::set_error("dvc_se_gate_static: hipGetSymbolAddress(scratch): %s", hipGetErrorString(e));
      return DVC_EHIP;
    }
    p = static_cast<float*>(sym);
    g_scratch_ptr[device].store(p, std::memory_order_release);
  }
  *out = p;
  return DVC_OK;
}
}  // namespace

extern "C" int dvc_se_gate_static(const DvcSrc* x, int32_t batch, int32_t h, int32_t w, const float* fc1,
                                  const float* fc2, int32_t hidden, float* gate, void* stream) {
  DVC_REQUIRE(batch >= 1 && h >= 1 && w >= 1, "dvc_se_gate_static: bad dims");
  DVC_CHECK_SRC(x, h, w, "dvc_se_gate_static");
  DVC_REQUIRE(x->channels <= 256 && hidden >= 1 && hidden <= 64, "dvc_se_gate_static: channels<=256, hidden<=64 required");
  DVC_REQUIRE((int64_t)batch * x->channels <= SE_STATIC_MAX_PLANES,
              "dvc_se_gate_static: batch * channels = %lld exceeds the built-in scratch (%d planes); use dvc_se_gate "
              "with caller-provided scratch", (long long)batch * x->channels, SE_STATIC_MAX_PLANES);
  DVC_REQUIRE(fc1 && fc2 && gate, "dvc_se_gate_static: null pointer");
  float* partial = nullptr;
  if (int rc = static_scratch(&partial)) return rc;
  // every (n, c, slice) partial sum this call reads is written by its first kernel: nothing to clear
  launch_plane_partial_sums(x, batch, h, w, partial, (hipStream_t)stream);
  int rc = dvc::check_launch("plane_partial_sum_kernel");
  if (rc) return rc;
  hipLaunchKernelGGL(se_fc_kernel, dim3(batch), dim3(256), 0, (hipStream_t)stream, (const float*)partial, x->channels,
                     hidden, 1.0f / ((float)h * (float)w), fc1, fc2, gate);
  return dvc::check_launch("se_fc_kernel");
}

extern "C" int dvc_copy(const DvcSrc* src, const DvcDst* out, int32_t batch, int32_t h, int32_t w, void* stream) {
  DVC_REQUIRE(batch >= 1 && h >= 1 && w >= 1, "dvc_copy: bad dims");
  DVC_CHECK_SRC(src, h, w, "dvc_copy");
  DVC_CHECK_DST(out, (int64_t)h * w, "dvc_copy");
  const long long total = (long long)batch * src->channels * h * w;
  if (((long long)h * w) % 4 == 0 && total / 4 < (1LL << 31) && reinterpret_cast<uintptr_t>(src->ptr) % 16 == 0 &&
      reinterpret_cast<uintptr_t>(out->ptr) % 16 == 0 && src->batch_stride % 4 == 0 && src->chan_stride % 4 == 0 &&
      out->batch_stride % 4 == 0 && out->chan_stride % 4 == 0) {
    hipLaunchKernelGGL(copy4_kernel, dim3(stream_grid(total / 4, 256)), dim3(256), 0, (hipStream_t)stream, mk(src), mk(out),
                       (unsigned)src->channels, (unsigned)((long long)h * w / 4), (unsigned)(total / 4));
    return dvc::check_launch("copy4_kernel");
  }
  hipLaunchKernelGGL(copy_kernel, dim3(stream_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, mk(src), mk(out),
                     batch, h, w);
  return dvc::check_launch("copy_kernel");
}

extern "C" int dvc_clamp(const DvcSrc* src, const DvcDst* out, int32_t batch, int32_t h, int32_t w, float lo,
                         float hi, void* stream) {
  DVC_REQUIRE(batch >= 1 && h >= 1 && w >= 1, "dvc_clamp: bad dims");
  DVC_CHECK_SRC(src, h, w, "dvc_clamp");
  DVC_CHECK_DST(out, (int64_t)h * w, "dvc_clamp");
  const long long total = (long long)batch * src->channels * h * w;
  hipLaunchKernelGGL(clamp_kernel, dim3(stream_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, mk(src), mk(out),
                     batch, h, w, lo, hi);
  return dvc::check_launch("clamp_kernel");
}
