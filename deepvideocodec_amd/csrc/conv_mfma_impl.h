// Kernel template and tile dispatch of the MFMA convolution (see conv_mfma.hip for the description).
// Included by the per-shape translation units conv_mfma_*.hip, which instantiate disjoint sets of
// kernel variants so that the library builds in parallel.
#pragma once
#include <atomic>
#include <type_traits>
#include <utility>

#include "dvc_common.h"
#include "conv_args.h"

namespace {

using dvc::ConvArgs;

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// 16-bit operand element of the reduced-operand modes: bf16 (modes 1, 2) or fp16 (mode 3).  The fp16
// conversion saturates instead of overflowing to infinity.
template <bool F16> struct Half16 { using T = __bf16; using V8 = bf16x8; };
template <> struct Half16<true> { using T = _Float16; using V8 = f16x8; };
template <bool F16>
__device__ __forceinline__ typename Half16<F16>::T to_half16(float v) {
  if constexpr (F16) return (_Float16)fminf(fmaxf(v, -65504.f), 65504.f);
  else return (__bf16)v;
}
__device__ __forceinline__ f32x16 mfma16(bf16x8 a, bf16x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mfma16(f16x8 a, f16x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}


// f(integral_constant<int, 0>) ... f(integral_constant<int, N-1>): a loop whose index is a constant expression in the
// body (register arrays indexed by it are never addressed dynamically, whatever the unroller decides)
template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  static_for_impl(f, std::make_integer_sequence<int, N>{});
}

// One LDS dword at (per-lane byte address + immediate): issues the read and returns at once -- the value is valid
// only behind an s_waitcnt lgkmcnt that covers it, which the caller places (the compiler does not track this read).
template <int OFFSET_BYTES>
__device__ __forceinline__ float lds_read_imm(unsigned addr) {
  static_assert(OFFSET_BYTES >= 0 && OFFSET_BYTES < 65536, "DS instructions carry a 16-bit offset");
  float v;
  asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFFSET_BYTES));
  return v;
}

template <int OFFSET_BYTES>
__device__ __forceinline__ void lds_write_imm(unsigned addr, float v) {
  static_assert(OFFSET_BYTES >= 0 && OFFSET_BYTES < 65536, "DS instructions carry a 16-bit offset");
  asm volatile("ds_write_b32 %0, %1 offset:%2" : : "v"(addr), "v"(v), "n"(OFFSET_BYTES) : "memory");
}

// input channels staged per K-chunk (even)
constexpr int chunk_of(int ks, int s) { return dvc::conv_chunk_of(ks, s); }
constexpr int cout_block_of(int cout) { return cout > 32 ? 64 : 32; }
// floats per packed weight slab: CK * k*k rows of CB output channels, padded to whole 1 KiB
// pieces (one global_load_lds wave-instruction moves 64 lanes x 16 B)
constexpr int slab_floats(int ks, int s, int cb) { return dvc::conv_slab_floats(ks, s, cb); }

// bf16 compute mode: K step of the MFMA is 16 input channels, held by a lane as 8 consecutive
// channels (one "octet", 16 bytes).  A slab is [tap][octet 0/1][co][8 channels] in bf16, its size
// given here in 4-byte units like the fp32 slabs (padded to whole 1 KiB pieces).
constexpr int BF_CK = 16;
constexpr int slab_floats_bf16(int ks, int cb) { return (ks * ks * 2 * cb * 4 + 255) / 256 * 256; }
// 3x3 / 7x7 stride 1 in the reduced-operand modes: one slab per (chunk, tap row), [part][kx][octet][co][8 x 16 bit],
// padded to whole 4 KiB (Geo::STEPPED / Geo::WSTEP)
constexpr bool reduced_stepped(int ks, int s) { return ks != 1 && s == 1; }
constexpr int step_slab_floats(int ks, int cb, int parts) { return (parts * ks * 2 * cb * 4 + 1023) / 1024 * 1024; }

// HALF: a 32-channel workgroup (MT = 1) working on one half of a 64-channel packed weight block,
// used to double the number of workgroups of small launches whose weights are packed 64 wide.
// BF: 0 = exact fp32 MFMA.  1 = operands rounded to bf16 on v_mfma_f32_32x32x16_bf16, fp32 accumulate.
// 2 = every operand split into two bf16 parts, hi = bf16(v), lo = bf16(v - hi) (16 mantissa bits), and
// a product taken as hi*hi + hi*lo + lo*hi on the same MFMA: three bf16 MFMAs per 16 channels instead of
// eight fp32 ones, relative error of a product <= ~2^-16.  With two parts the operand tiles are twice as
// large and the LDS buffers are single (the next chunk is staged after the MFMA phase, not during it).
// 3 = the same split with fp16 parts on v_mfma_f32_32x32x16_f16: hi = fp16(v), lo = fp16(v - hi) carry 22
// significant bits (absolute floor 2^-25 from the fp16 subnormals, which the MFMA keeps), so a product is
// within ~2^-21 of the fp32 one; operands saturate at +-65504 (hi) + 65504 (lo).
// DY = 2 (7x7 stride 1, 16 output channels, MT = 1): the row-pair form.  The 32 MFMA rows are the 16 filters and
// the same 16 filters moved one tap row down, so one accumulator tile holds two output rows of 32 pixels; the tap
// grid is 8 x 7 and a wave's tiles are two input rows apart.  8 x 7 MFMA steps per two output rows instead of
// 2 x 7 x 7, and the sums are those of the ordinary form (the extra terms are exact zeros).
template <int KS, int S, int MT, int NT, bool HALF = false, int BF = 0, int DY = 1>
struct Geo {
  static_assert(!HALF || MT == 1, "half blocks are 32 channels");
  static_assert(DY == 1 || (DY == 2 && KS == 7 && S == 1 && MT == 1 && !HALF && BF == 0), "row pairs: 7x7 stride 1 fp32");
  static constexpr int PARTS = BF >= 2 ? 2 : 1;
  static constexpr bool F16 = BF == 3;
  // Reduced-operand 3x3 / 7x7 stride-1 kernels walk K in STEPS of (16-channel chunk, one tap row): the weights of a
  // step are one small slab that is DMA'd into a ring of NW slots one or two steps before it is read, so no wave ever
  // waits for a weight slab it has just asked for, and the 49-tap weights of a 7x7 chunk (200 KiB in two parts)
  // never have to be resident at once.  The converted input tile of a chunk stays for its KS steps; 3x3 keeps two of
  // them, so the next chunk is converted and written while the current one multiplies.
  static constexpr bool STEPPED = BF != 0 && KS != 1 && S == 1;
  // Wave-specialised form of the largest split-operand 3x3 tile (16 rows x 32 pixels x 64 channels): a workgroup of
  // EIGHT waves, one per CU.  Waves 0..3 only multiply (four rows of 32 pixels each, 128 accumulator registers, no
  // staging registers); waves 4..7 only stage: they request the next chunk's weight slabs (DMA) and input tile,
  // split the inputs into their 16-bit parts and write them to the other LDS buffer while the first four waves
  // multiply the current chunk.  Each SIMD then holds one multiplying wave that has the matrix pipe to itself and
  // one staging wave whose vector work runs beside it, with ONE barrier per chunk: the symmetric 4-wave form at two
  // workgroups per CU kept the pipe about half busy (both workgroups of a CU are outside their multiply phases a
  // quarter of the time, collide inside them another quarter, and a barrier per tap row waits for the slowest of
  // four waves each time).
  static constexpr bool WSP = BF >= 2 && KS == 3 && S == 1 && MT == 2 && NT == 4 && !HALF;
  static constexpr int THREADS = WSP ? 512 : 256;
#ifndef DVC_STEP_RING
#define DVC_STEP_RING 2
#endif
  static constexpr int NW = KS == 3 ? DVC_STEP_RING : 2;
  static constexpr int NBUF = WSP ? 2 : STEPPED ? (KS == 3 && DVC_STEP_RING == 3 ? 2 : 1) : (PARTS == 2 ? 1 : 2);   // LDS buffers of the input tile
  static constexpr int CK = BF ? BF_CK : chunk_of(KS, S);
  static constexpr int TH = 4 * NT, TW = 32, CB = 32 * MT, PAD = KS / 2;
  static constexpr int SL = (KS == 1) ? 1 : S;   // pixel stride inside the LDS tile
  static constexpr int KSY = KS + DY - 1;        // tap rows
  static constexpr int SLY = DY > 1 ? DY : SL;   // input rows between two tile rows
  static constexpr int IH = (TH - 1) * SLY + KSY;
  static constexpr int IW = (TW - 1) * SL + KS;
  static constexpr int CHS = IH * IW;            // floats per staged channel
  static constexpr int TAPS = KSY * KS;
  static constexpr int KCH = CK * TAPS;          // k values per chunk
  static constexpr int CBP = HALF ? 64 : CB;     // output channels per packed weight row
  static constexpr int SLAB = DY > 1 ? dvc::conv_rowpair_slab_floats()
                              : BF ? PARTS * slab_floats_bf16(KS, CBP) : slab_floats(KS, S, CBP);   // in 4-byte units
  // fp32: [channel][element]; bf16: [octet][element][8 channels] = 8 * CHS 4-byte units
  static constexpr int IN_BUF = ((BF ? 8 * PARTS : CK) * CHS + 3) / 4 * 4;   // keep 16-byte alignment of what follows
  static constexpr int IN_PT = (CHS + 255) / 256;         // input elements per thread per channel
  static constexpr int GS = (KS == 1) ? S : 1;            // global step between staged pixels
  // step slab [part][kx][octet][co][8 x 16 bit] in 4-byte units, padded to whole 4 KiB (each of the four waves
  // moves the same number of 1 KiB pieces: the counted waits below rely on it)
  static constexpr int WSTEP = (PARTS * KS * 2 * CBP * 4 + 1023) / 1024 * 1024;
  static constexpr int W_UNITS = WSP ? 2 * KS * WSTEP : STEPPED ? NW * WSTEP : NBUF * SLAB;   // LDS of the weights, 4-byte units
  static constexpr int STAGE_BYTES = (W_UNITS + NBUF * IN_BUF) * 4;
  // workgroups per CU the LDS allows (the register budget follows it)
  // (three per CU only where a wave's accumulators leave room in a 168-register budget: 64 of them at most)
  static constexpr int MIN_WAVES = (!STEPPED || WSP) ? 2 : STAGE_BYTES > 80 * 1024 ? 1 : (STAGE_BYTES + 320 > 53 * 1024 || MT * NT > 4) ? 2 : 3;
  // four wave-private 32 x 36 transposition tiles + the four waves' per-channel sums (plane-sum output)
  static constexpr int EPI_BYTES = 4 * 32 * 36 * 4 + 4 * 64 * 4;
  static constexpr int MAIN_BYTES = STAGE_BYTES > EPI_BYTES ? STAGE_BYTES : EPI_BYTES;
  static constexpr int LDS_BYTES = MAIN_BYTES + 64 * 4;   // + this workgroup's (up to 64) bias values
};

// Diagnostic build only (-DDVC_CONV_STAMPS, tools/conv_stamps.py): wave 0 of every workgroup records the
// clock at its phase boundaries into a buffer no other code reads.  The shipped library has no stamps.
#ifdef DVC_CONV_STAMPS
__device__ unsigned long long* g_conv_stamps = nullptr;
#define DVC_STAMP(k)                                                                                          \
  do {                                                                                                        \
    if (g_conv_stamps != nullptr && threadIdx.x == 0) {                                                       \
      const unsigned long long wg_ = stamp_wg; \
      g_conv_stamps[wg_ * 16 + (k)] = __builtin_amdgcn_s_memrealtime();                                       \
      if ((k) == 0) {                                                                                         \
        g_conv_stamps[wg_ * 16 + 4] = __builtin_amdgcn_s_memtime();                                           \
        g_conv_stamps[wg_ * 16 + 6] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) |         \
                                     ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32); \
      }                                                                                                       \
      if ((k) == 3) g_conv_stamps[wg_ * 16 + 5] = __builtin_amdgcn_s_memtime();                               \
    }                                                                                                         \
  } while (0)
#else
#define DVC_STAMP(k)
#endif
// -DDVC_STAMP_CHUNKS (with DVC_CONV_STAMPS): shader-clock stamps of the phases of K chunks 2 and 3 in slots 8..15
// instead of the epilogue's per-tile stamps
#if defined(DVC_CONV_STAMPS) && defined(DVC_STAMP_CHUNKS)
#define DVC_CSTAMP(cond, slot)                                                                                \
  do {                                                                                                        \
    if (g_conv_stamps != nullptr && threadIdx.x == 0 && (cond)) {                                             \
      const unsigned long long wg_ = stamp_wg; \
      g_conv_stamps[wg_ * 16 + (slot)] = __builtin_amdgcn_s_memtime();                                        \
    }                                                                                                         \
  } while (0)
#define DVC_CSTAMP_T(cond, slot, thread)                                                                      \
  do {                                                                                                        \
    if (g_conv_stamps != nullptr && threadIdx.x == (thread) && (cond)) {                                      \
      const unsigned long long wg_ = stamp_wg; \
      g_conv_stamps[wg_ * 16 + (slot)] = __builtin_amdgcn_s_memtime();                                        \
    }                                                                                                         \
  } while (0)
#else
#define DVC_CSTAMP(cond, slot)
#define DVC_CSTAMP_T(cond, slot, thread)
#endif

// s_barrier that the compiler may not move LDS accesses across (the bare builtin is not a memory operation for the
// optimiser), without the wait for every outstanding vector-memory request that __syncthreads()'s fences add: the
// stepped K loop keeps weight-slab DMAs in flight across barriers and places its own counted waits.
__device__ __forceinline__ void lds_barrier() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

template <int KS, int S, int MT, int NT, bool HALF = false, int BF = 0, int DY = 1>
__global__ __launch_bounds__((Geo<KS, S, MT, NT, HALF, BF, DY>::THREADS), (Geo<KS, S, MT, NT, HALF, BF, DY>::MIN_WAVES))
void conv_mfma_f32_kernel(const ConvArgs a) {
  using G = Geo<KS, S, MT, NT, HALF, BF, DY>;
  constexpr int CBP = G::CBP;
  constexpr int CK = G::CK, TH = G::TH, TW = G::TW, CB = G::CB, PAD = G::PAD, SL = G::SL, IW = G::IW, SLY = G::SLY;
  constexpr int CHS = G::CHS, TAPS = G::TAPS, SLAB = G::SLAB, IN_BUF = G::IN_BUF, IN_PT = G::IN_PT, GS = G::GS;

  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int NBUF = G::NBUF, PARTS = G::PARTS;
  float* const w_s = lds;                   // [NBUF][SLAB]   (first: 1 KiB-aligned pieces for the LDS DMA)
  float* const in_s = lds + G::W_UNITS;     // [NBUF][IN_BUF]
  float* const bias_s = lds + G::MAIN_BYTES / 4;   // [CB] bias of this workgroup's channels (0 past Cout / without bias)

  const int tid = G::WSP ? (int)(threadIdx.x & 255) : (int)threadIdx.x;   // index among the 256 staging (or multiplying) lanes
  const int lane = tid & 63, half = lane >> 5, l32 = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);               // 0..3 within the wave's role
  const bool stager = G::WSP && __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 8)) != 0;   // waves 4..7 of a wave-specialised workgroup
#ifdef DVC_CONV_STAMPS
  // record index of the stamps: the workgroup, or (persistent form) workgroups x tiles walked so far + workgroup
  unsigned long long stamp_wg = blockIdx.x + (unsigned long long)gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
#define DVC_STAMP_NEXT_TILE() (stamp_wg += gridDim.x)
#else
#define DVC_STAMP_NEXT_TILE()
#endif
  if constexpr (!G::WSP) DVC_STAMP(0);
  // ---- per-tile state.  A workgroup computes one tile (set once from its block index), or -- wave-specialised
  // form -- walks a list of them (set again per tile, by each wave role at its own pace).
  int tile = 0, n = 0, wblock = 0, cb0 = 0, oy0 = 0, ox0 = 0;
  const int nchunks = (a.cin + CK - 1) / CK;
  // byte offset of each staged element inside its channel plane; padding elements get an offset
  // past every plane, for which the buffer load returns 0 (hardware range check) -- zero padding
  // and the ragged last channel chunk cost no select instructions.
  unsigned in_boff[IN_PT];
  const unsigned plane_bytes = 4u * (unsigned)(a.h * a.w);
  const float* wslabs = a.wpacked;
  // The (up to 4) concatenated sources, hoisted out of the kernel-argument segment once (plain
  // scalars, not arrays: resolving a channel to its plane inside the K loop must cost neither
  // dependent scalar loads nor scratch traffic).
  // (unused sources have channels == 0 and a null pointer: they are never selected)
  const float *sp0 = nullptr, *sp1 = nullptr, *sp2 = nullptr, *sp3 = nullptr;
  // global -> LDS DMA of weight slabs is issued as buffer loads (MUBUF ... lds), not global_load_lds: the compiler
  // treats a FLAT-family instruction that touches LDS as able to return out of order with the ds_reads and then
  // waits for lgkmcnt(0) before every MFMA group, which exposes the LDS latency of the operand reads just issued
  // for the next tap (10 % of the K loop for a wave alone on its SIMD).
  __amdgpu_buffer_rsrc_t wrsrc;
  auto set_tile = [&](int tile_, int by, int bz) __attribute__((always_inline)) {
    tile = tile_;
    const int ty = tile / a.tiles_x, tx = tile - ty * a.tiles_x;
    oy0 = ty * TH;
    ox0 = tx * TW;
    wblock = HALF ? (by >> 1) : by;   // packed weight block
    cb0 = HALF ? wblock * 64 + (by & 1) * 32 : wblock * CB;
    n = bz;
    // per-thread staging descriptors (identical for every channel and chunk)
    const int gy0 = (KS == 1) ? oy0 * S : oy0 * (DY > 1 ? DY : S) - PAD;   // (row pairs: oy0 counts pairs)
    const int gx0 = (KS == 1) ? ox0 * S : ox0 * S - PAD;
#pragma unroll
    for (int j = 0; j < IN_PT; ++j) {
      const int i = tid + 256 * j;
      const int iy = i / IW, ix = i - iy * IW;
      const int gy = gy0 + iy * GS, gx = gx0 + ix * GS;
      const bool ok = (i < CHS) && gy >= 0 && gy < a.h && gx >= 0 && gx < a.w;
      in_boff[j] = ok ? 4u * (unsigned)(gy * a.w + gx) : 0x80000000u;
    }
    wslabs = a.wpacked + (long long)wblock * nchunks * (G::STEPPED ? KS * G::WSTEP : SLAB);
    wrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(wslabs), 0, (unsigned)(4u * (unsigned)nchunks * (unsigned)(G::STEPPED ? KS * G::WSTEP : SLAB)), 0x00020000);
    sp0 = a.src[0].ptr + (long long)n * a.src[0].batch_stride;
    sp1 = a.src[1].ptr + (long long)n * a.src[1].batch_stride;
    sp2 = a.src[2].ptr + (long long)n * a.src[2].batch_stride;
    sp3 = a.src[3].ptr + (long long)n * a.src[3].batch_stride;
  };
  if constexpr (!G::WSP) {
    // Workgroups are handed to the 8 XCDs round-robin in launch order and every XCD has its own L2.
    // Tiles are therefore renumbered so that each XCD works through one contiguous row-major range
    // of them: the halo columns and rows shared by neighbouring tiles are then fetched into one L2
    // once instead of into up to three.  (Any launch order gives correct results; this is locality.)
    const int nt = gridDim.x;
    const int s = (int)((blockIdx.y + blockIdx.z * gridDim.y) * gridDim.x) & 7;   // XCD phase of this grid row
    const int u = blockIdx.x + s, k = u & 7;
    // launch slots of this row that land on XCDs < k: F(s + nt) - F(s), F(m) = (m / 8) k + min(m % 8, k)
    const int m1 = s + nt;
    const int before = (m1 >> 3) * k + min(m1 & 7, k) - min(s, k);
    const int first_u = s + ((k - s) & 7);
    set_tile(before + ((u - first_u) >> 3), (int)blockIdx.y, (int)blockIdx.z);
    if (HALF && cb0 >= a.cout) return;   // upper half of a ragged last block: nothing to compute (workgroup-uniform)
    if constexpr (DY > 1) {   // rows 16..31 are the 16 channels again
      if (tid < CB) bias_s[tid] = a.bias != nullptr ? a.bias[tid & 15] : 0.f;
    } else {
      if (tid < CB) bias_s[tid] = (a.bias != nullptr && cb0 + tid < a.cout) ? a.bias[cb0 + tid] : 0.f;
    }
  }

  float rin[CK][IN_PT];

  // global -> LDS DMA of one weight slab: SLAB/256 pieces of 1 KiB, round-robin over the 4 waves.
  auto dma_weights = [&](int chunk, int buf) __attribute__((always_inline)) {
    float* dst = w_s + buf * SLAB;
    const unsigned soff = 4u * (unsigned)chunk * (unsigned)SLAB;
#pragma unroll
    for (int p = 0; p < (SLAB / 256 + 3) / 4; ++p) {
      const int piece = wave + 4 * p;
      if (piece < SLAB / 256)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, (__attribute__((address_space(3))) void*)(dst + piece * 256), 16,
                                             16u * (unsigned)lane, soff + 1024u * (unsigned)piece, 0, 0);
    }
  };

  const long long cs0 = a.src[0].chan_stride, cs1 = a.src[1].chan_stride;
  const long long cs2 = a.src[2].chan_stride, cs3 = a.src[3].chan_stride;
  const int e0 = a.src[0].channels;   // first channel past source 0, 1, 2
  const int e1 = e0 + a.src[1].channels;
  const int e2 = e1 + a.src[2].channels;
  const int cin = a.cin;
  const bool single_src = a.n_src == 1;
  const bool in_act = a.in_act != 0;
  const float in_slope = a.in_slope;

  const bool soff_ok = a.soff_ok != 0;

  // One K-chunk of input, global -> registers.  Common case (the chunk lies inside one source and
  // inside cin): one buffer descriptor per chunk on the source's sample, the channel as the scalar
  // offset operand of the load (not part of the range check, which keeps working per plane through
  // the per-lane offsets) -- a couple of scalar instructions per channel and no branches, so the
  // loads can be scheduled between the MFMAs of the current chunk.  Otherwise (a chunk straddling two
  // concatenated sources, or running past the last channel): one descriptor per channel.
  // State of a chunk's input fetch: the buffer descriptor base and channel stride of the common case.
  struct Fetch { unsigned lo, hi, cs4, soff0; bool fast; };
  // Starts the fetch of chunk c0: works out the common-case state, or (chunk straddling sources / running past
  // the last channel) issues all of the chunk's loads with one descriptor per channel and returns fast = false.
  // (`issue(descriptor, scalar offset, channel index constant)` requests one channel of the chunk for this lane)
  auto prefetch_begin_with = [&](int c0, auto&& issue) __attribute__((always_inline)) -> Fetch {
    const int cl = c0 + CK - 1;
    const unsigned long long b0 = reinterpret_cast<unsigned long long>(sp0);
    const unsigned long long b1 = reinterpret_cast<unsigned long long>(sp1);
    const unsigned long long b2 = reinterpret_cast<unsigned long long>(sp2), b3 = reinterpret_cast<unsigned long long>(sp3);
    // plane lookup over the concatenated sources as mask arithmetic (select chains on these scalars
    // get turned into a private-memory lookup table by the compiler); the telescoping sums work
    // because g2 implies g1 implies g0
    const bool g0 = c0 >= e0, g1 = c0 >= e1, g2 = c0 >= e2;
    const bool fast = soff_ok && cl < cin && (cl >= e0) == g0 && (cl >= e1) == g1 && (cl >= e2) == g2;
    if (fast) {   // workgroup-uniform
      const unsigned long long m0 = 0ull - (unsigned long long)g0, m1 = 0ull - (unsigned long long)g1;
      const unsigned long long m2 = 0ull - (unsigned long long)g2;
      const unsigned long long base = b0 + ((b1 - b0) & m0) + ((b2 - b1) & m1) + ((b3 - b2) & m2);
      const long long cs = cs0 + ((cs1 - cs0) & (long long)m0) + ((cs2 - cs1) & (long long)m1) + ((cs3 - cs2) & (long long)m2);
      const int first = (e0 & (int)m0) + ((e1 - e0) & (int)m1) + ((e2 - e1) & (int)m2);
      Fetch f;
      f.lo = __builtin_amdgcn_readfirstlane((unsigned)base);
      f.hi = __builtin_amdgcn_readfirstlane((unsigned)(base >> 32));
      f.cs4 = __builtin_amdgcn_readfirstlane(4u * (unsigned)cs);
      f.soff0 = __builtin_amdgcn_readfirstlane((unsigned)(c0 - first) * f.cs4);
      f.fast = true;
      return f;
    }
    static_for<CK>([&](auto cc) __attribute__((always_inline)) {
      constexpr int c = decltype(cc)::value;
      const int ci = c0 + c;
      const bool have = ci < cin;
      unsigned long long pl;
      if (single_src) {   // workgroup-uniform
        pl = have ? b0 + 4ull * (unsigned long long)((long long)ci * cs0) : b0;
      } else {
        const unsigned long long m0 = 0ull - (unsigned long long)(ci >= e0);
        const unsigned long long m1 = 0ull - (unsigned long long)(ci >= e1);
        const unsigned long long m2 = 0ull - (unsigned long long)(ci >= e2);
        const unsigned long long base = b0 + ((b1 - b0) & m0) + ((b2 - b1) & m1) + ((b3 - b2) & m2);
        const long long cs = cs0 + ((cs1 - cs0) & (long long)m0) + ((cs2 - cs1) & (long long)m1) + ((cs3 - cs2) & (long long)m2);
        const int first = (e0 & (int)m0) + ((e1 - e0) & (int)m1) + ((e2 - e1) & (int)m2);
        pl = have ? base + 4ull * (unsigned long long)((long long)(ci - first) * cs) : b0;
      }
      // wave-uniform buffer descriptor (made provably so for the compiler): base = plane,
      // size = H*W*4 bytes, or 0 past the last channel so that every load returns 0
      const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)pl);
      const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(pl >> 32));
      const unsigned nbytes = __builtin_amdgcn_readfirstlane(have ? plane_bytes : 0u);
      const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
          reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo), 0, nbytes, 0x00020000);
      issue(rsrc, 0u, cc);
    });
    return Fetch{0u, 0u, 0u, 0u, false};
  };
  auto issue_dwords = [&](const __amdgpu_buffer_rsrc_t rsrc, unsigned soff, auto cc) __attribute__((always_inline)) {
    constexpr int c = decltype(cc)::value;
#pragma unroll
    for (int j = 0; j < IN_PT; ++j)
      rin[c][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, in_boff[j], soff, 0));
  };
  auto prefetch_begin = [&](int c0) __attribute__((always_inline)) -> Fetch { return prefetch_begin_with(c0, issue_dwords); };
  // the common case's descriptor of channel c of the chunk and the scalar offset that goes with it
  auto channel_rsrc = [&](const Fetch& f, unsigned soff) __attribute__((always_inline)) {
    // the range check compares the per-lane offset with (records - scalar offset): records end
    // with this channel's plane
    return __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<void*>(((unsigned long long)f.hi << 32) | f.lo), 0, soff + plane_bytes, 0x00020000);
  };
  // Common case (the chunk lies inside one source and inside cin): one buffer descriptor on the source's sample,
  // the channel as the scalar offset operand of the load (not part of the range check, which keeps working per
  // plane through the per-lane offsets) -- a couple of scalar instructions per channel and no branches, so a
  // channel's loads fit between two MFMAs of the current chunk.
  auto prefetch_channel = [&](const Fetch& f, int c) __attribute__((always_inline)) {
    const unsigned soff = f.soff0 + (unsigned)c * f.cs4;
    const __amdgpu_buffer_rsrc_t rsrc = channel_rsrc(f, soff);
#pragma unroll
    for (int j = 0; j < IN_PT; ++j)
      rin[c][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, in_boff[j], soff, 0));
  };
  // One K-chunk of input, global -> registers, all at once.
  auto prefetch_input = [&](int c0) __attribute__((always_inline)) {
    const Fetch f = prefetch_begin(c0);
    if (f.fast) {   // workgroup-uniform
      static_for<CK>([&](auto cc) __attribute__((always_inline)) { prefetch_channel(f, decltype(cc)::value); });
    }
  };

  // registers -> LDS, one channel of the chunk (fp32 layout [channel][element]), as explicit ds_write_b32 with
  // immediate offsets off the per-lane byte address `dst_addr` of the buffer: no address arithmetic (VALU) between
  // the MFMAs this is issued behind.  The compiler does not track these writes: the chunk's closing s_waitcnt
  // covers them (lgkmcnt(0) before the barrier).
  auto commit_channel = [&](unsigned dst_addr, auto cc) __attribute__((always_inline)) {
    constexpr int c = decltype(cc)::value;
    if (in_act) {   // wave-uniform
#pragma unroll
      for (int j = 0; j < IN_PT; ++j) rin[c][j] = dvc::leaky(rin[c][j], in_slope);
    }
    static_for<IN_PT>([&](auto jc) __attribute__((always_inline)) {
      constexpr int j = decltype(jc)::value;
      if (256 * (j + 1) <= CHS || tid + 256 * j < CHS)   // only the last slice of a channel can run past its end
        lds_write_imm<4 * (c * CHS + 256 * j)>(dst_addr, rin[c][j]);
    });
  };
  auto commit_input = [&](int buf) __attribute__((always_inline)) {
    float* dst = in_s + buf * IN_BUF + tid;
    if (in_act) {   // wave-uniform
#pragma unroll
      for (int c = 0; c < CK; ++c)
#pragma unroll
        for (int j = 0; j < IN_PT; ++j) rin[c][j] = dvc::leaky(rin[c][j], in_slope);
    }
    if constexpr (BF) {
      // element i of the tile gets its 16 channels as two 16-byte octets: [octet][i][8 x bf16]
      using V8 = typename Half16<G::F16>::V8;
      V8* const d8 = reinterpret_cast<V8*>(in_s + buf * IN_BUF) + tid;
#pragma unroll
      for (int j = 0; j < IN_PT; ++j)
        if (256 * (j + 1) <= CHS || tid + 256 * j < CHS) {
#pragma unroll
          for (int o = 0; o < 2; ++o) {
            V8 hi, lo;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              hi[e] = to_half16<G::F16>(rin[8 * o + e][j]);
              if constexpr (PARTS == 2) lo[e] = to_half16<G::F16>(rin[8 * o + e][j] - (float)hi[e]);
            }
            d8[o * CHS + 256 * j] = hi;
            if constexpr (PARTS == 2) d8[(2 + o) * CHS + 256 * j] = lo;   // [part][octet][element]
          }
        }
    } else {
#pragma unroll
      for (int c = 0; c < CK; ++c) {
#pragma unroll
        for (int j = 0; j < IN_PT; ++j)   // only the last slice of a channel can run past its end
          if (256 * (j + 1) <= CHS || tid + 256 * j < CHS) dst[c * CHS + 256 * j] = rin[c][j];
      }
    }
  };

  f32x16 acc[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

  const int b_lane = half * CHS + (wave * NT * SLY) * IW + l32 * SL;
  const int a_lane = (half * TAPS) * CBP + l32 + (HALF ? 32 * (int)(blockIdx.y & 1) : 0);
  // bf16 mode, in 16-byte units: A [tap][octet][co], B [octet][element]
  const int a_lane8 = half * CBP + l32 + (HALF ? 32 * (int)(blockIdx.y & 1) : 0);
  const int b_lane8 = half * CHS + (wave * NT * SL) * IW + l32 * SL;

  // ---- the tile's epilogue (run once the K loop is done; `epi`: LDS the transposition tiles may use)
  // (the lane coordinates are parameters so that the persistent form can hand in copies the optimiser cannot trace:
  // the per-lane store offsets would otherwise be hoisted out of the tile loop and held across the K loop)
  auto epilogue = [&](float* const epi, const int lane, const int half, const int l32) __attribute__((always_inline)) {
  DVC_STAMP(2);
#ifdef DVC_EXP_NO_EPILOGUE   // experiment: how long the launch takes with no epilogue at all (wrong results)
  if (a.h < 0) {
    float sacc = 0.f;
    for (int mt = 0; mt < MT; ++mt) for (int nt = 0; nt < NT; ++nt) for (int r = 0; r < 16; ++r) sacc += acc[mt][nt][r];
    a.out[tid] = sacc;
  }
  return;
#endif
  // ---- epilogue: bias, activation, residual(s), (pixel-shuffle) store.
  // Accumulator register r of tile mt holds output channel cb0 + 32 mt + (r&3) + 8 (r>>2) + 4 half.
  // Addresses are formed as (wave-uniform channel base pointer) + (32-bit per-lane offset) so the
  // 128 stores of a lane cost no 64-bit vector address arithmetic.
  const bool has_res = a.res != nullptr, has_res2 = a.res2 != nullptr, has_scale = a.res_scale != nullptr;
  float* const out_n = a.out + (long long)n * a.obs;
  const float* const res_n = has_res ? a.res + (long long)n * a.rbs : nullptr;
  const float* const res2_n = has_res2 ? a.res2 + (long long)n * a.r2bs : nullptr;

  if constexpr (DY > 1) {
    // Row pairs (launched only for sizes the tile divides, 16-byte aligned rows, no residual / shuffle): rows
    // 0..15 of an accumulator tile are output row 2p, rows 16..31 output row 2p + 1 of channels 0..15.  The same
    // trip through the wave's LDS tile as below; piece q of a lane = channel 8 (q & 1) + sub of row 2p + (q >> 1).
    constexpr int RS = 36;
    float* const tl = epi + wave * (32 * RS);
    const int sub = lane >> 3, px4 = 4 * (lane & 7);
    const unsigned o_lane = (unsigned)sub * (unsigned)a.ocs + (unsigned)px4;
    const int lds_lane = sub * RS + px4;
    const float out_slope = a.out_slope;
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    f32x4 cur[4];
    float bias_r[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) bias_r[r] = bias_s[4 * half + (r & 3) + 8 * (r >> 2)];
    // (a constant-index loop: with a run-time index the four pieces end up in private memory)
    static_for<NT + 1>([&](auto ic) __attribute__((always_inline)) {
      constexpr int i = decltype(ic)::value;
      if constexpr (i < NT) {
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          const f32x2 v = f32x2{acc[0][i][r], acc[0][i][r + 1]} + f32x2{bias_r[r], bias_r[r + 1]};
          const f32x2 sv = v * out_slope;
          tl[((r & 3) + 8 * (r >> 2) + 4 * half) * RS + l32] = fmaxf(v.x, sv.x);
          tl[(((r + 1) & 3) + 8 * (r >> 2) + 4 * half) * RS + l32] = fmaxf(v.y, sv.y);
        }
      }
      if constexpr (i > 0) {
        const long long row0 = 2LL * (oy0 + wave * NT + (i - 1));
#pragma unroll
        for (int q = 0; q < 4; ++q)
          *reinterpret_cast<f32x4*>(out_n + (long long)(8 * (q & 1)) * a.ocs + (row0 + (q >> 1)) * a.wo + ox0 + o_lane) = cur[q];
      }
      if constexpr (i < NT) {
#pragma unroll
        for (int q = 0; q < 4; ++q) cur[q] = *reinterpret_cast<const f32x4*>(tl + 8 * q * RS + lds_lane);
      }
    });
    DVC_STAMP(3);
    return;
  }
  if (a.vec4) {
    // Fast path (everything 16-byte aligned).  Store instructions, not bytes, are what the epilogue
    // pays for, so each 32 channel x 32 pixel accumulator tile is turned through a wave-private LDS
    // tile ([channel][pixel], row stride RS) and leaves as 16-byte stores: 4 per lane instead of 16.
    // All waves are past the K loop's last barrier, so the staging buffers are free; a wave's LDS
    // instructions execute in order, so its reads see its own writes without a barrier.
    // The code is instantiated per residual configuration so that a tile is straight-line code: the
    // residual loads of a tile are issued before its trip through LDS and are waited for only at the add.
    constexpr int RS = 36;
    static_assert(G::LDS_BYTES >= 4 * 32 * RS * 4, "epilogue tiles do not fit the staging buffers");
    float* const tl = epi + wave * (32 * RS);
    auto run = [&](auto res_c, auto scale_c, auto res2_c, auto ps_c) {
      constexpr bool RES = decltype(res_c)::value, SCALE = decltype(scale_c)::value;
      constexpr bool RES2 = decltype(res2_c)::value, PS = decltype(ps_c)::value;
      // plain: 8 lanes x 16 B per channel row, quad q covers channels 8q + sub
      // shuffle: conv channel co -> shuffled channel co>>2 at (2 oy + ((co>>1)&1), 2 ox + (co&1)): an output
      // row interleaves two adjacent conv channels; 16 lanes x 16 B per output row, a lane takes 2 pixels of
      // both channels = 4 outputs; quad q covers output rows 4q + row of the tile (row = (channel, dy))
      const int sub = lane >> 3, px4 = 4 * (lane & 7);
      const int row = lane >> 4, px2 = 2 * (lane & 15);
      const long long wo2 = 2LL * a.wo;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const int cu0 = cb0 + mt * 32;
        float bias_r[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) bias_r[r] = bias_s[mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const int oy = oy0 + wave * NT + nt;
          if (oy >= a.ho) continue;   // wave-uniform
          // where the four 16-byte pieces of this lane go (stored-tensor channel, pixel offset)
          bool ok[4];
          int chs[4];
          long long pix[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            if constexpr (!PS) {
              chs[q] = cu0 + 8 * q + sub;
              ok[q] = chs[q] < a.cout && ox0 + px4 < a.wo;
              pix[q] = (long long)oy * a.wo + ox0 + px4;
            } else {
              const int rr = 4 * q + row, c2l = rr >> 1, dy = rr & 1;
              chs[q] = (cu0 >> 2) + c2l;
              ok[q] = cu0 + 4 * c2l + 2 * dy < a.cout && ox0 + px2 < a.wo;
              pix[q] = (2LL * oy + dy) * wo2 + 2 * (ox0 + px2);
            }
          }
          float4 t[4], u[4];
          float sc[4];
          if constexpr (RES) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {   // lanes without a piece read the sample's first 16 bytes
              t[q] = *reinterpret_cast<const float4*>(res_n + (ok[q] ? chs[q] * a.rcs + pix[q] : 0LL));
              if constexpr (RES2) u[q] = *reinterpret_cast<const float4*>(res2_n + (ok[q] ? chs[q] * a.r2cs + pix[q] : 0LL));
              if constexpr (SCALE) sc[q] = a.res_scale[n * a.c_final + (ok[q] ? chs[q] : 0)];
            }
          }
#pragma unroll
          for (int r = 0; r < 16; ++r)
            tl[((r & 3) + 8 * (r >> 2) + 4 * half) * RS + l32] = dvc::leaky(acc[mt][nt][r] + bias_r[r], a.out_slope);
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            float4 v;
            if constexpr (!PS) {
              v = *reinterpret_cast<const float4*>(tl + (8 * q + sub) * RS + px4);
            } else {
              const int rr = 4 * q + row, chl = 4 * (rr >> 1) + 2 * (rr & 1);
              const float2 e = *reinterpret_cast<const float2*>(tl + chl * RS + px2);
              const float2 o = *reinterpret_cast<const float2*>(tl + (chl + 1) * RS + px2);
              v = make_float4(e.x, o.x, e.y, o.y);
            }
            if constexpr (RES) {
              if constexpr (SCALE) {
                // product and sum rounded separately, like the unfused reference graph (x * gate, then +)
                v.x = __fadd_rn(v.x, __fmul_rn(t[q].x, sc[q])); v.y = __fadd_rn(v.y, __fmul_rn(t[q].y, sc[q]));
                v.z = __fadd_rn(v.z, __fmul_rn(t[q].z, sc[q])); v.w = __fadd_rn(v.w, __fmul_rn(t[q].w, sc[q]));
              } else {
                v.x += t[q].x; v.y += t[q].y; v.z += t[q].z; v.w += t[q].w;
              }
              if constexpr (RES2) { v.x += u[q].x; v.y += u[q].y; v.z += u[q].z; v.w += u[q].w; }
            }
            if (ok[q]) *reinterpret_cast<float4*>(out_n + chs[q] * a.ocs + pix[q]) = v;
          }
#ifndef DVC_STAMP_CHUNKS
          DVC_STAMP(8 + ((mt * NT + nt) & 7));
#endif
        }
      }
    };
    // The common case -- no PixelShuffle, the whole 32*MT x TH x 32 tile inside the tensor -- as one
    // branch-free software pipeline over the MT*NT accumulator tiles of the wave: tile i is written to the
    // wave's LDS tile and read back transposed, and while those reads are in flight tile i-1 gets its
    // residuals added and is stored (LDS executes a wave's accesses in order, so the single tile buffer
    // needs no waits between tiles).  Addresses are a wave-uniform base per (tile, channel octet) plus one
    // 32-bit per-lane offset computed once; the bias comes from LDS.
    auto run_full = [&](auto res_c, auto scale_c, auto res2_c, auto ps_c, auto sums_c) {
      constexpr bool RES = decltype(res_c)::value, SCALE = decltype(scale_c)::value;
      constexpr bool RES2 = decltype(res2_c)::value, PS = decltype(ps_c)::value;
      // SUMS: also emit, per output channel, the sum of the values this workgroup stores (the first stage of a
      // deterministic two-stage plane sum: the squeeze-excite gate needs the plane means of this very tensor, and
      // a separate pass would read all of it back from HBM)
      constexpr bool SUMS = decltype(sums_c)::value;
      static_assert(!SUMS || (!RES && !PS), "plane sums: plain stores without residual");
      float psum[MT][4];
      if constexpr (SUMS) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int q = 0; q < 4; ++q) psum[mt][q] = 0.f;
      }
      // plain: 8 lanes x 16 B per channel row; piece q of a lane = channel 8q + sub, pixels px4 .. px4+3.
      // shuffle: conv channel co -> stored channel co>>2 at (2 oy + ((co>>1)&1), 2 ox + (co&1)); 16 lanes x 16 B
      // per stored row; piece q = stored channel 2q + (row>>1), row parity row&1, conv pixels px2, px2+1 of the
      // two conv channels 8q + 4(row>>1) + 2(row&1) + {0, 1}.
      const int sub = lane >> 3, px4 = 4 * (lane & 7);
      const int row = lane >> 4, px2 = 2 * (lane & 15);
      const long long wo2 = 2LL * a.wo;
      const unsigned o_lane = PS ? (unsigned)(row >> 1) * (unsigned)a.ocs + (unsigned)((row & 1) * wo2) + 2u * px2
                                 : (unsigned)sub * (unsigned)a.ocs + (unsigned)px4;
      const unsigned r_lane = PS ? (unsigned)(row >> 1) * (unsigned)a.rcs + (unsigned)((row & 1) * wo2) + 2u * px2
                                 : (unsigned)sub * (unsigned)a.rcs + (unsigned)px4;
      const unsigned r2_lane = PS ? (unsigned)(row >> 1) * (unsigned)a.r2cs + (unsigned)((row & 1) * wo2) + 2u * px2
                                  : (unsigned)sub * (unsigned)a.r2cs + (unsigned)px4;
      const int lds_lane = PS ? (4 * (row >> 1) + 2 * (row & 1)) * RS + px2 : sub * RS + px4;
      const int cq = PS ? 2 : 8;             // stored channels per piece index q
      const int c0s = PS ? (cb0 >> 2) : cb0;   // first stored channel of the workgroup
      const int cms = PS ? 8 : 32;           // stored channels per 32-channel accumulator group
      const float out_slope = a.out_slope;
      // residual operands are requested D tiles ahead of their use.  (Looking two tiles ahead in the
      // one-workgroup-per-CU form, which has no second workgroup to hide the wait behind, needs 16 more registers
      // than its 256 hold -- it spilled; that form does request the second residual ahead, which the others load at
      // its use)
      constexpr int D = 1;
      float4 cur[4], t[D][4];
      float4 t2[(RES2 && G::WSP) ? D : 1][4];
      float bias_r[16];
      float sc[MT][4];
      if constexpr (SCALE) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int q = 0; q < 4; ++q)
            sc[mt][q] = a.res_scale[n * a.c_final + c0s + mt * cms + cq * q + (PS ? (row >> 1) : sub)];
      }
      auto tile_pix = [&](int nt) -> long long {   // wave-uniform pixel offset of the tile inside a stored plane
        const long long oy = oy0 + wave * NT + nt;
        return PS ? 2 * oy * wo2 + 2LL * ox0 : oy * a.wo + ox0;
      };
      auto finish = [&](int mt, int nt, int slot) {   // residual adds + stores of a tile whose transposed values are in `cur`
        const long long pix = tile_pix(nt);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float4 v = cur[q];
          const long long ch = c0s + mt * cms + cq * q;   // wave-uniform stored channel of this piece (lane 0's)
          if constexpr (RES) {
            if constexpr (SCALE) {
              // product and sum rounded separately, like the unfused reference graph (x * gate, then +)
              const float g = sc[mt][q];
              v.x = __fadd_rn(v.x, __fmul_rn(t[slot][q].x, g)); v.y = __fadd_rn(v.y, __fmul_rn(t[slot][q].y, g));
              v.z = __fadd_rn(v.z, __fmul_rn(t[slot][q].z, g)); v.w = __fadd_rn(v.w, __fmul_rn(t[slot][q].w, g));
            } else {
              v.x += t[slot][q].x; v.y += t[slot][q].y; v.z += t[slot][q].z; v.w += t[slot][q].w;
            }
            if constexpr (RES2) {
              // (rare: loaded where it is used, to stay within the register budget -- except in the one-workgroup-per-CU
              // form, where no other workgroup hides the wait and the operand is requested a tile ahead like `t`)
              float4 u;
              if constexpr (G::WSP) u = t2[slot][q];
              else u = *reinterpret_cast<const float4*>(res2_n + ch * a.r2cs + pix + r2_lane);
              v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
            }
          }
          *reinterpret_cast<float4*>(out_n + ch * a.ocs + pix + o_lane) = v;
          if constexpr (SUMS) psum[mt][q] += (v.x + v.y) + (v.z + v.w);
        }
      };
      auto request_residuals = [&](int j) {   // tile j's residual operands into slot j % D
        const int mt = j / NT, nt = j - mt * NT;
        const long long pix = tile_pix(nt);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          t[j % D][q] = *reinterpret_cast<const float4*>(res_n + (long long)(c0s + mt * cms + cq * q) * a.rcs + pix + r_lane);
          if constexpr (RES2 && G::WSP)
            t2[j % D][q] = *reinterpret_cast<const float4*>(res2_n + (long long)(c0s + mt * cms + cq * q) * a.r2cs + pix + r2_lane);
        }
      };
      if constexpr (RES) {
#pragma unroll
        for (int j = 0; j < D - 1; ++j) request_residuals(j);
      }
#pragma unroll
      for (int i = 0; i <= MT * NT; ++i) {
        if (i < MT * NT) {
          const int mt = i / NT, nt = i - mt * NT;
          if (nt == 0) {   // one batch of LDS reads per 32-channel group (not one dependent read per value)
#pragma unroll
            for (int r = 0; r < 16; ++r) bias_r[r] = bias_s[mt * 32 + 4 * half + (r & 3) + 8 * (r >> 2)];
          }
#pragma unroll
          for (int r = 0; r < 16; r += 2) {   // two values per instruction (v_pk_add_f32, v_pk_mul_f32)
            const f32x2 v = f32x2{acc[mt][nt][r], acc[mt][nt][r + 1]} + f32x2{bias_r[r], bias_r[r + 1]};
            const f32x2 sv = v * out_slope;
            // LeakyReLU / ReLU / identity (slope 1) as max(v, v * slope): equal to the select for 0 <= slope <= 1
            tl[((r & 3) + 8 * (r >> 2) + 4 * half) * RS + l32] = fmaxf(v.x, sv.x);
            tl[(((r + 1) & 3) + 8 * (r >> 2) + 4 * half) * RS + l32] = fmaxf(v.y, sv.y);
          }
        }
        if (i > 0) finish((i - 1) / NT, (i - 1) % NT, (i - 1) % D);
        if (i < MT * NT) {
          // read tile i back transposed; consumed one iteration later, behind the next tile's arithmetic
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            if constexpr (!PS) {
              cur[q] = *reinterpret_cast<const float4*>(tl + 8 * q * RS + lds_lane);
            } else {
              const float2 e = *reinterpret_cast<const float2*>(tl + 8 * q * RS + lds_lane);
              const float2 o = *reinterpret_cast<const float2*>(tl + (8 * q + 1) * RS + lds_lane);
              cur[q] = make_float4(e.x, o.x, e.y, o.y);
            }
          }
        }
        if constexpr (RES) {   // into the slot the tile just stored has left (D = 1: this tile's own, consumed one iteration later)
          if (i + D - 1 < MT * NT) request_residuals(i + D - 1);
        }
      }
      if constexpr (SUMS) {
        // a lane's psum[mt][q] covers 4 pixels x NT rows of channel 32 mt + 8 q + sub; the 8 lanes of a channel
        // are adjacent: fold them, then the four waves through LDS, always in the same order
        float* const red = epi + 4 * (32 * RS);   // [wave][CB], behind the transposition tiles
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            float s = psum[mt][q];
            s += __shfl_xor(s, 1, 64);
            s += __shfl_xor(s, 2, 64);
            s += __shfl_xor(s, 4, 64);
            if ((lane & 7) == 0) red[wave * CB + mt * 32 + 8 * q + sub] = s;
          }
        __syncthreads();
        if (tid < CB)
          a.plane_sums[((long long)n * a.cout + cb0 + tid) * a.sum_slots + tile] =
              (red[tid] + red[CB + tid]) + (red[2 * CB + tid] + red[3 * CB + tid]);
      }
    };
    using T = std::true_type;
    using F = std::false_type;
    // (slopes outside [0, 1] never occur in the DMC graph; they take the general formulation below)
    const bool full = cb0 + CB <= a.cout && ox0 + TW <= a.wo && oy0 + TH <= a.ho &&
                      a.out_slope >= 0.f && a.out_slope <= 1.f;
    auto by_shuffle = [&](auto res_c, auto scale_c, auto res2_c) {
      if (full) {
        if (a.ps) run_full(res_c, scale_c, res2_c, T{}, F{});
        else run_full(res_c, scale_c, res2_c, F{}, F{});
      } else if (a.ps) {
        run(res_c, scale_c, res2_c, T{});
      } else {
        run(res_c, scale_c, res2_c, F{});
      }
    };
    if (!has_res) {
      // (plane sums are requested only for launches made of whole tiles -- conv_plane_sum_slots on the host)
      if constexpr (KS == 3 && S == 1 && BF == 0 && !HALF && DY == 1) {
        if (a.plane_sums != nullptr && full && !a.ps) {
          run_full(F{}, F{}, F{}, F{}, T{});
          DVC_STAMP(3);
          return;
        }
      }
      by_shuffle(F{}, F{}, F{});
    }
    else if (has_scale) { if (has_res2) by_shuffle(T{}, T{}, T{}); else by_shuffle(T{}, T{}, F{}); }
    else { if (has_res2) by_shuffle(T{}, F{}, T{}); else by_shuffle(T{}, F{}, F{}); }
    DVC_STAMP(3);
    return;
  }

  // General path: one 4-byte store per accumulator register.
  const int ox = ox0 + l32;
  if (ox >= a.wo) return;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int cu0 = cb0 + mt * 32;   // wave-uniform first channel of the tile
    float bias_r[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = cu0 + (r & 3) + 8 * (r >> 2) + 4 * half;
      bias_r[r] = (a.bias != nullptr && co < a.cout) ? a.bias[co] : 0.f;
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int oy = oy0 + wave * NT + nt;
      if (oy >= a.ho) continue;
      float rv[16], rv2[16];
      if (!a.ps) {
        const unsigned pix = (unsigned)(oy * a.wo + ox);
        const unsigned o_lane = 4u * half * (unsigned)a.ocs + pix;
        const unsigned r_lane = 4u * half * (unsigned)a.rcs + pix;
        const unsigned r2_lane = 4u * half * (unsigned)a.r2cs + pix;
        if (has_res) {   // issue all residual loads of the tile before using any
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int cu = cu0 + (r & 3) + 8 * (r >> 2);
            const bool ok = cu + 4 * half < a.cout;
            // channels past Cout do not exist in the residual: such registers read channel 0, offset 0
            const int cu_ld = cu < a.cout ? cu : 0;   // wave-uniform
            rv[r] = (res_n + (long long)cu_ld * a.rcs)[ok ? r_lane : 0u];
            if (has_scale) rv[r] = __fmul_rn(rv[r], a.res_scale[n * a.c_final + (ok ? cu + 4 * half : 0)]);
            rv2[r] = has_res2 ? (res2_n + (long long)cu_ld * a.r2cs)[ok ? r2_lane : 0u] : 0.f;
          }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int cu = cu0 + (r & 3) + 8 * (r >> 2);
          if (cu + 4 * half >= a.cout) continue;
          float v = acc[mt][nt][r] + bias_r[r];
          v = dvc::leaky(v, a.out_slope);   // slope 1 when the layer has no output activation
          if (has_res) {
            v = __fadd_rn(v, rv[r]);
            if (has_res2) v = __fadd_rn(v, rv2[r]);
          }
          (out_n + (long long)cu * a.ocs)[o_lane] = v;
        }
      } else {
        // conv channel co -> shuffled channel co>>2 at (2 oy + ((co>>1)&1), 2 ox + (co&1)); registers
        // r, r+1 (r even) are the two horizontal neighbours, stored as one float2.  With
        // co = cu + 4 half: shuffled channel (cu>>2) + half, row parity (cu>>1)&1.
        const unsigned wo2 = 2u * (unsigned)a.wo;
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          const int cu = cu0 + (r & 3) + 8 * (r >> 2);   // even
          const bool ok = cu + 4 * half < a.cout;
          const int c2u = cu >> 2, dy = (cu >> 1) & 1;
          const unsigned pix = (unsigned)(2 * oy + dy) * wo2 + 2u * (unsigned)ox;
          float v0 = acc[mt][nt][r] + bias_r[r], v1 = acc[mt][nt][r + 1] + bias_r[r + 1];
          v0 = dvc::leaky(v0, a.out_slope);
          v1 = dvc::leaky(v1, a.out_slope);
          if (has_res && ok) {
            const float2 t = *reinterpret_cast<const float2*>(res_n + (long long)c2u * a.rcs + (half * (unsigned)a.rcs + pix));
            const float sc = has_scale ? a.res_scale[n * a.c_final + c2u + half] : 1.f;
            v0 = __fadd_rn(v0, has_scale ? __fmul_rn(t.x, sc) : t.x);
            v1 = __fadd_rn(v1, has_scale ? __fmul_rn(t.y, sc) : t.y);
            if (has_res2) {
              const float2 u = *reinterpret_cast<const float2*>(res2_n + (long long)c2u * a.r2cs + (half * (unsigned)a.r2cs + pix));
              v0 += u.x;
              v1 += u.y;
            }
          }
          if (ok)
            *reinterpret_cast<float2*>(out_n + (long long)c2u * a.ocs + (half * (unsigned)a.ocs + pix)) = make_float2(v0, v1);
        }
      }
    }
  }
  };

  if constexpr (G::WSP) {
    // ---- wave-specialised K loop (see Geo::WSP): one barrier per chunk, taken by all eight waves.
    // Persistent: the launch is one workgroup per CU, and workgroup b walks the work items (sample, channel block,
    // tile; the tile fastest) first, first + stride, ... of the contiguous eighth of the launch that belongs to its
    // XCD (workgroups go to the XCDs round-robin).  Chunks are numbered through the whole walk: chunk g lives in
    // buffer pair g & 1, and while the multiplying waves are in the last chunk of a tile the staging waves are
    // already bringing in the first chunk of the next -- a tile's fill is hidden behind its predecessor.  The
    // epilogue's transposition tiles use the weight buffer of the chunk just finished; one more barrier per tile keeps
    // the staging waves from refilling it (they are one chunk ahead and would otherwise reach it next).
    constexpr int WSTEP = G::WSTEP, WCH = KS * WSTEP;   // a chunk's three tap-row slabs are contiguous in the packed image
    static_assert(WCH % 512 == 0, "whole 1 KiB pieces per staging wave");
    static_assert(WCH * 4 >= 4 * 32 * 36 * 4, "epilogue tiles inside one weight buffer");
    using V8 = typename Half16<G::F16>::V8;
    const int stride = (int)(gridDim.x >> 3);
    const int first = (int)((long long)(blockIdx.x & 7) * a.total_items / 8) + (int)(blockIdx.x >> 3);
    const int last = (int)((long long)((blockIdx.x & 7) + 1) * a.total_items / 8);
    if (first >= last) return;   // (workgroup-uniform, before any barrier)
    const int n_total = ((last - first + stride - 1) / stride) * nchunks;   // chunks of this workgroup's walk
    auto enter = [&](int item) __attribute__((always_inline)) {
      const int per_sample = a.tiles_xy * a.cblocks;
      const int bz = item / per_sample, rem = item - bz * per_sample;
      const int by = rem / a.tiles_xy;
      set_tile(rem - by * a.tiles_xy, by, bz);
    };
    if (stager) {
      // The four staging waves work as two pairs that swap roles every chunk interval.  In the interval in which
      // the multiplying waves are in chunk v, the pair (v & 1) STREAMS: it requests the weight slabs of chunk v + 1
      // (DMA into the buffer chunk v - 1 has left) and the input tile of chunk v + 2 (into registers: no buffer is
      // free for it yet), and the other pair CONVERTS: it splits the input of chunk v + 1, which it requested one
      // interval earlier, into its 16-bit parts and writes it to LDS.  A wave that requests memory spends the
      // interval stalled at the request queue (a CU sustains about 10 bytes per cycle of such requests, whatever the
      // request size or the number of CUs at work -- measured), so requests and conversion only overlap when they
      // come from different waves; in lockstep they add up to more than the chunk's 216 MFMAs.
      const int pair = wave >> 1, ptid = tid & 127;
      constexpr int PT2 = (CHS + 127) / 128;       // tile elements per lane of a pair and channel
      float rin2[CK][PT2];
      unsigned boff2[PT2];
      __amdgpu_buffer_rsrc_t wrsrc2;
      const int per_sample = a.tiles_xy * a.cblocks;
      // tile state of the input cursor (this pair's next chunk to request) and of the weight cursor
      auto enter_input = [&](int item) __attribute__((always_inline)) {
        const int bz = item / per_sample, rem = item - bz * per_sample;
        const int t = rem % a.tiles_xy;
        const int ty = t / a.tiles_x, tx = t - ty * a.tiles_x;
        const int gy0 = ty * TH - PAD, gx0 = tx * TW - PAD;
#pragma unroll
        for (int j = 0; j < PT2; ++j) {
          const int i = ptid + 128 * j;
          const int iy = i / IW, ix = i - iy * IW;
          const int gy = gy0 + iy, gx = gx0 + ix;
          const bool ok = (i < CHS) && gy >= 0 && gy < a.h && gx >= 0 && gx < a.w;
          boff2[j] = ok ? 4u * (unsigned)(gy * a.w + gx) : 0x80000000u;
        }
        sp0 = a.src[0].ptr + (long long)bz * a.src[0].batch_stride;
        sp1 = a.src[1].ptr + (long long)bz * a.src[1].batch_stride;
        sp2 = a.src[2].ptr + (long long)bz * a.src[2].batch_stride;
        sp3 = a.src[3].ptr + (long long)bz * a.src[3].batch_stride;
      };
      auto enter_weights = [&](int item) __attribute__((always_inline)) {
        const int wb = (item % per_sample) / a.tiles_xy;
        wrsrc2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.wpacked + (long long)wb * nchunks * WCH), 0,
                                                   (unsigned)(4u * (unsigned)nchunks * (unsigned)WCH), 0x00020000);
      };
      // a cursor over every second chunk of the walk: (item, chunk of the item)
      auto advance = [&](int& item, int& c) __attribute__((always_inline)) -> bool {   // true: the item changed
        c += 2;
        bool moved = false;
        while (c >= nchunks) { c -= nchunks; item += stride; moved = true; }
        return moved;
      };
      auto issue_input = [&](const __amdgpu_buffer_rsrc_t rsrc, unsigned so, auto cc) __attribute__((always_inline)) {
        constexpr int c = decltype(cc)::value;
#pragma unroll
        for (int j = 0; j < PT2; ++j)
          rin2[c][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, boff2[j], so, 0));
      };
      int in_item = first, in_c = pair - 2;      // chunks pair, pair + 2, ... of the walk
      int w_item = first, w_c = (pair ^ 1) - 2;  // chunks pair ^ 1, ...
      if (advance(in_item, in_c) || true) { if (in_item < last) enter_input(in_item); }
      if (advance(w_item, w_c) || true) { if (w_item < last) enter_weights(w_item); }
      int cv = 0;   // chunk of the tile the multiplying waves are in
      for (int v = -2;; ++v) {
        if ((v & 1) == pair) {
          DVC_CSTAMP_T(cv == 2, 11, 256);   // (diagnostic build; four chunks per tile: wave 4 streams in chunks 0 and 2)
          if (v >= -1 && w_item < last) {   // weight slabs of chunk v + 1: 1 KiB pieces, round-robin over the pair
            float* dst = w_s + ((v + 1) & 1) * WCH;
            const unsigned soff = 4u * (unsigned)w_c * (unsigned)WCH;
#pragma unroll
            for (int p = 0; p < WCH / 512; ++p) {   // WCH / 256 pieces, half of them per wave of the pair
              const int piece = (wave & 1) + 2 * p;
              __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc2, (__attribute__((address_space(3))) void*)(dst + piece * 256), 16,
                                                       16u * (unsigned)lane, soff + 1024u * (unsigned)piece, 0, 0);
            }
            if (advance(w_item, w_c) && w_item < last) enter_weights(w_item);
          }
          if (in_item < last) {             // input of chunk v + 2
            const Fetch f = prefetch_begin_with(in_c * CK, issue_input);
            if (f.fast) {
              static_for<CK>([&](auto cc) __attribute__((always_inline)) {
                const unsigned so = f.soff0 + (unsigned)decltype(cc)::value * f.cs4;
                issue_input(channel_rsrc(f, so), so, cc);
              });
            }
            __builtin_amdgcn_s_waitcnt(0xCF7F);   /* vmcnt(63): the input requests just made may stay in flight; the DMA pieces before them have landed */
          } else {
            __builtin_amdgcn_s_waitcnt(0x0F70);   /* vmcnt(0) */
          }
          DVC_CSTAMP_T(cv == 2, 12, 256);
        } else {
          DVC_CSTAMP_T(cv == 1, 14, 256);
          if (v >= -1 && in_item < last) {  // the input of chunk v + 1, requested one interval ago
            V8* const d8 = reinterpret_cast<V8*>(in_s + ((v + 1) & 1) * IN_BUF) + ptid;
#pragma unroll
            for (int j = 0; j < PT2; ++j)
              if (128 * (j + 1) <= CHS || ptid + 128 * j < CHS) {
#pragma unroll
                for (int o = 0; o < 2; ++o) {
                  V8 hi, lo;
#pragma unroll
                  for (int e = 0; e < 8; ++e) {
                    float x = rin2[8 * o + e][j];
                    if (in_act) x = dvc::leaky(x, in_slope);
                    hi[e] = to_half16<G::F16>(x);
                    lo[e] = to_half16<G::F16>(x - (float)hi[e]);
                  }
                  d8[o * CHS + 128 * j] = hi;
                  d8[(2 + o) * CHS + 128 * j] = lo;   // [part][octet][element]
                }
              }
            if (advance(in_item, in_c) && in_item < last) enter_input(in_item);
            __builtin_amdgcn_s_waitcnt(0xC07F);   /* lgkmcnt(0): the LDS writes have landed */
          }
          DVC_CSTAMP_T(cv == 1, 15, 256);
        }
        lds_barrier();
        DVC_CSTAMP_T(cv == 1, 13, 256);
        if (v >= 0) {
          if (++cv == nchunks) {
            cv = 0;
            lds_barrier();   // the multiplying waves are done with the epilogue's LDS tiles
            DVC_STAMP_NEXT_TILE();
            if (v + 1 >= n_total) break;
          }
        }
      }
      return;
    }
    constexpr int APART = KS * 2 * CBP, BPART = 2 * CHS;   // second operand part, in 16-byte units
    int g = 0;
    lds_barrier();   // (the staging waves' two start-up intervals)
    lds_barrier();
    for (int item = first; item < last; item += stride) {
      DVC_STAMP(0);
      enter(item);
      DVC_STAMP(1);
      // (read by the epilogue, nchunks barriers from here; the previous tile's epilogue is behind the tile barrier)
      if (tid < CB) bias_s[tid] = (a.bias != nullptr && cb0 + tid < a.cout) ? a.bias[cb0 + tid] : 0.f;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;
      for (int c = 0; c < nchunks; ++c, ++g) {
        const int cur = g & 1;
        const V8* const bbase = reinterpret_cast<const V8*>(in_s + cur * IN_BUF) + b_lane8;
        const V8* const wbase = reinterpret_cast<const V8*>(w_s + cur * WCH) + a_lane8;
        DVC_CSTAMP(c == 1, 8);
#pragma unroll
        for (int r = 0; r < KS; ++r) {
          const V8* const abase = wbase + r * (WSTEP / 4);
#pragma unroll
          for (int kx = 0; kx < KS; ++kx) {
            V8 ah[MT], al[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
              ah[mt] = abase[kx * 2 * CBP + mt * 32];
              al[mt] = abase[APART + kx * 2 * CBP + mt * 32];
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
              const V8 bh = bbase[(nt + r) * IW + kx];
              const V8 bl = bbase[BPART + (nt + r) * IW + kx];
#pragma unroll
              for (int mt = 0; mt < MT; ++mt) {   // the small cross terms first, then the leading one
                acc[mt][nt] = mfma16(al[mt], bh, acc[mt][nt]);
                acc[mt][nt] = mfma16(ah[mt], bl, acc[mt][nt]);
                acc[mt][nt] = mfma16(ah[mt], bh, acc[mt][nt]);
              }
            }
          }
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);   /* lgkmcnt(0): this wave is done reading the chunk's buffers */
        DVC_CSTAMP(c == 1, 9);
        lds_barrier();
        DVC_CSTAMP(c == 1, 10);
      }
      int lane_e = lane;
      asm volatile("" : "+v"(lane_e));
      epilogue(w_s + ((g - 1) & 1) * WCH, lane_e, lane_e >> 5, lane_e & 31);
      __builtin_amdgcn_s_waitcnt(0xC07F);   /* lgkmcnt(0): this wave's trips through the LDS tiles are over */
      lds_barrier();
      DVC_STAMP_NEXT_TILE();
    }
    return;
  } else if constexpr (G::STEPPED) {
    // ---- K loop in steps of (chunk, tap row); see Geo::STEPPED.  Per step: [last row of a chunk: convert the next
    // chunk's input (requested during row 0) into the other input buffer] -> request the slab NW-1 steps ahead ->
    // [row 0: request the next chunk's input] -> the row's MFMAs -> wait until the NEXT step's slab has landed (all
    // but this step's own requests: the counter retires in order) -> barrier.  Hazards: a ring slot is overwritten
    // NW-1 steps after its last read and an input buffer one chunk after its, each with a barrier in between.
    constexpr int R = KS, NW = G::NW, WSTEP = G::WSTEP, P = WSTEP / 1024, L = CK * IN_PT;
    static_assert(NW != 3 || P + L <= 63, "counted vmcnt exceeds the field");
    using V8 = typename Half16<G::F16>::V8;
    const int n_steps = nchunks * R;
    auto dma_step = [&](int step) __attribute__((always_inline)) {
      const int slot = step % NW;
      float* dst = w_s + slot * WSTEP;
      const unsigned soff = 4u * (unsigned)step * (unsigned)WSTEP;
#pragma unroll
      for (int p = 0; p < P; ++p) {
        const int piece = wave + 4 * p;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, (__attribute__((address_space(3))) void*)(dst + piece * 256), 16,
                                                 16u * (unsigned)lane, soff + 1024u * (unsigned)piece, 0, 0);
      }
    };
    // s_waitcnt immediates: vmcnt(n) lgkmcnt(0) (vmcnt is split over bits 3:0 and 15:14, expcnt left at 7)
    constexpr int kVm0Lgkm0 = 0x0070;
    constexpr int kVmP = (P & 0xF) | ((P >> 4) << 14) | 0x0070, kVmL = (L & 0xF) | ((L >> 4) << 14) | 0x0070;
    constexpr int kVmPL = ((P + L) & 0xF) | (((P + L) >> 4) << 14) | 0x0070;
    dma_step(0);
    if (NW == 3 && n_steps > 1) dma_step(1);
    prefetch_input(0);
    commit_input(0);
    __builtin_amdgcn_s_waitcnt(kVm0Lgkm0);
    lds_barrier();
    DVC_STAMP(1);
    constexpr int APART = KS * 2 * CBP, BPART = 2 * CHS;   // second operand part, in 16-byte units
    for (int c = 0; c < nchunks; ++c) {
      const int cur = NBUF == 2 ? (c & 1) : 0, nxt = NBUF == 2 ? (cur ^ 1) : 0;
      const bool more = c + 1 < nchunks;
      const V8* const bbase = reinterpret_cast<const V8*>(in_s + cur * IN_BUF) + b_lane8;
      static_for<R>([&](auto rc) __attribute__((always_inline)) {
        constexpr int r = decltype(rc)::value;
        const int step = c * R + r;
        if (NBUF == 2 && r == R - 1 && more) commit_input(nxt);
        const bool dma_now = step + NW - 1 < n_steps;
        if (dma_now) dma_step(step + NW - 1);
        const bool loads_now = r == 0 && more;
        if (loads_now) prefetch_input((c + 1) * CK);
        const V8* const abase = reinterpret_cast<const V8*>(w_s + (step % NW) * WSTEP) + a_lane8;
        if constexpr (r < 3) DVC_CSTAMP(c == 1, 8 + 2 * r);    // (diagnostic build: second chunk, before / after the row's MFMAs)
#pragma unroll
        for (int kx = 0; kx < KS; ++kx) {
          V8 ah[MT], al[MT];
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) {
            ah[mt] = abase[kx * 2 * CBP + mt * 32];
            if constexpr (PARTS == 2) al[mt] = abase[APART + kx * 2 * CBP + mt * 32];
          }
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            const V8 bh = bbase[(nt + r) * IW + kx];
            if constexpr (PARTS == 2) {
              const V8 bl = bbase[BPART + (nt + r) * IW + kx];
#pragma unroll
              for (int mt = 0; mt < MT; ++mt) {   // the small cross terms first, then the leading one
                acc[mt][nt] = mfma16(al[mt], bh, acc[mt][nt]);
                acc[mt][nt] = mfma16(ah[mt], bl, acc[mt][nt]);
                acc[mt][nt] = mfma16(ah[mt], bh, acc[mt][nt]);
              }
            } else {
#pragma unroll
              for (int mt = 0; mt < MT; ++mt) acc[mt][nt] = mfma16(ah[mt], bh, acc[mt][nt]);
            }
          }
        }
        if constexpr (r < 3) DVC_CSTAMP(c == 1, 9 + 2 * r);
        if (NBUF == 1 && r == R - 1 && more) {   // one input buffer: everyone is done with it, then it is rewritten
          __builtin_amdgcn_s_waitcnt(0xC07F);   /* lgkmcnt(0) */
          lds_barrier();
          commit_input(0);
        }
        // the next step's slab: requested one step ago (ring of three: everything this step asked for may stay in
        // flight) or at the top of this step (ring of two)
        if (NW == 3) {
          if (dma_now && loads_now) __builtin_amdgcn_s_waitcnt(kVmPL);
          else if (dma_now) __builtin_amdgcn_s_waitcnt(kVmP);
          else if (loads_now) __builtin_amdgcn_s_waitcnt(kVmL);
          else __builtin_amdgcn_s_waitcnt(kVm0Lgkm0);
        } else {
          __builtin_amdgcn_s_waitcnt(kVm0Lgkm0);
        }
        if constexpr (r == 2) DVC_CSTAMP(c == 1, 14);
        lds_barrier();
        if constexpr (r == 2) DVC_CSTAMP(c == 1, 15);
      });
    }
  } else {
  // ---- prologue: chunk 0 into buffer 0
  dma_weights(0, 0);
  prefetch_input(0);
  commit_input(0);
  __builtin_amdgcn_s_waitcnt(0x0F70);   /* vmcnt(0): the DMA pieces have landed (a form the compiler's wait-count tracking sees) */
  __syncthreads();
  DVC_STAMP(1);

  for (int k = 0; k < nchunks; ++k) {
    const int cur = NBUF == 2 ? (k & 1) : 0, nxt = NBUF == 2 ? (cur ^ 1) : 0;
    DVC_CSTAMP(k == 2, 8);
    DVC_CSTAMP(k == 3, 13);
    const bool more = k + 1 < nchunks;
    // fp32 mode with enough tap groups per chunk: the next chunk's loads are issued one channel per tap group
    // behind the first CK groups' MFMAs and committed to LDS one channel per group behind the last CK groups'
    // (a workgroup that has its CU's matrix pipes to itself would otherwise leave them idle for that long)
    constexpr int T_GROUPS = (CK / 2) * TAPS;
    constexpr bool SPREAD = BF == 0 && T_GROUPS >= 2 * CK && MT * NT >= 2;
    // (with fewer than four MFMAs per tap group a chunk is over before its loads have come back: an early wait
    // inside the loop would hold up the MFMAs behind it, so those variants commit after the loop as before)
    constexpr bool SPREAD_COMMIT = SPREAD && NBUF == 2 && MT * NT >= 4;
    Fetch nextf{0u, 0u, 0u, 0u, false};
    auto start_next_chunk = [&]() __attribute__((always_inline)) {   // chunk k+1 in flight during the MFMA phase
      if (more) {
        if (NBUF == 2) dma_weights(k + 1, nxt);
        if constexpr (SPREAD) nextf = prefetch_begin((k + 1) * CK);
        else prefetch_input((k + 1) * CK);
      }
    };
    if constexpr (!SPREAD) start_next_chunk();   // (spread: behind the first MFMA of the chunk, see below)
    unsigned commit_addr = (unsigned)(size_t)(__attribute__((address_space(3))) float*)(in_s + nxt * IN_BUF + tid);
    asm volatile("" : "+v"(commit_addr));   // one register for the whole chunk, not re-derived at every use
    DVC_CSTAMP(k == 2, 9);
    if constexpr (BF) {
      using V8 = typename Half16<G::F16>::V8;
      const V8* bbase = reinterpret_cast<const V8*>(in_s + cur * IN_BUF) + b_lane8;
      const V8* abase = reinterpret_cast<const V8*>(w_s + cur * SLAB) + a_lane8;
      if constexpr (PARTS == 1) {
        V8 av[2][MT], bv[2][NT];
        auto fetch = [&](int tap, int slot) {
          const int ky = tap / KS, kx = tap - ky * KS;
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) av[slot][mt] = abase[tap * 2 * CBP + mt * 32];
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) bv[slot][nt] = bbase[(nt * SL + ky) * IW + kx];
        };
        fetch(0, 0);
#pragma unroll
        for (int t = 0; t < TAPS; ++t) {
          if (t + 1 < TAPS) fetch(t + 1, (t + 1) & 1);
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
              acc[mt][nt] = mfma16(av[t & 1][mt], bv[t & 1][nt], acc[mt][nt]);
        }
      } else {
        // split operands: A parts [part][tap][octet][co], B parts [part][octet][element]; the small
        // cross terms are accumulated before the leading one
        constexpr int APART = TAPS * 2 * CBP, BPART = 2 * CHS;
#pragma unroll
        for (int t = 0; t < TAPS; ++t) {
          const int ky = t / KS, kx = t - ky * KS;
          V8 ah[MT], al[MT];
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) {
            ah[mt] = abase[t * 2 * CBP + mt * 32];
            al[mt] = abase[APART + t * 2 * CBP + mt * 32];
          }
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            const V8 bh = bbase[(nt * SL + ky) * IW + kx];
            const V8 bl = bbase[BPART + (nt * SL + ky) * IW + kx];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
              acc[mt][nt] = mfma16(al[mt], bh, acc[mt][nt]);
              acc[mt][nt] = mfma16(ah[mt], bl, acc[mt][nt]);
              acc[mt][nt] = mfma16(ah[mt], bh, acc[mt][nt]);
            }
          }
        }
      }
    } else {
      // Operand reads as explicit ds_read_b32 with 16-bit immediate offsets off two per-lane base addresses, and
      // explicit wait counts.  Left to the compiler the reads are paired into ds_read2_b32, whose 8-bit offsets
      // need a fresh base register every few taps -- and a VALU instruction between two MFMA groups of one wave
      // waits for the matrix pipe to drain: about 40 of every 512 cycles when the wave has its SIMD to itself.
      // The compiler does not know these reads are asynchronous: every value is consumed only behind the
      // s_waitcnt that covers it, and the sched_barriers keep the MFMAs on their side of that wait.
      const unsigned b_addr = (unsigned)(size_t)(__attribute__((address_space(3))) const float*)(in_s + cur * IN_BUF + b_lane);
      const unsigned a_addr = (unsigned)(size_t)(__attribute__((address_space(3))) const float*)(w_s + cur * SLAB + a_lane);
      constexpr int T = (CK / 2) * TAPS;
      float av[2][MT], bv[2][NT];
      auto fetch = [&](auto tc) __attribute__((always_inline)) {
        constexpr int t = decltype(tc)::value, slot = t & 1;
        constexpr int cp = t / TAPS, tap = t - cp * TAPS, ky = tap / KS, kx = tap - ky * KS;
        static_for<MT>([&](auto mc) __attribute__((always_inline)) {
          constexpr int mt = decltype(mc)::value;
          av[slot][mt] = lds_read_imm<4 * ((2 * cp * TAPS + tap) * CBP + mt * 32)>(a_addr);
        });
        static_for<NT>([&](auto nc) __attribute__((always_inline)) {
          constexpr int nt = decltype(nc)::value;
          bv[slot][nt] = lds_read_imm<4 * (2 * cp * CHS + (nt * SLY + ky) * IW + kx)>(b_addr);
        });
      };
      static_assert(4 * (CK * TAPS * CBP + 32) < 65536 && 4 * (CK * CHS) < 65536, "operand offsets exceed the DS offset field");
      constexpr int kWaitAllButNext = 0xC07F | ((MT + NT) << 8);   // s_waitcnt lgkmcnt(MT + NT): the reads just issued may stay in flight
      constexpr int kWaitAll = 0xC07F;                             // s_waitcnt lgkmcnt(0)
      if constexpr (MT * NT == 1) {
        // one MFMA per tap: nothing to hide a wait behind, the compiler's own pairing and scheduling of the reads
        // is the faster form here
        const float* bbase = in_s + cur * IN_BUF + b_lane;
        const float* abase = w_s + cur * SLAB + a_lane;
        auto fetch1 = [&](int t, int slot) __attribute__((always_inline)) {
          const int cp = t / TAPS, tap = t - cp * TAPS, ky = tap / KS, kx = tap - ky * KS;
          av[slot][0] = abase[(2 * cp * TAPS + tap) * CBP];
          bv[slot][0] = bbase[2 * cp * CHS + ky * IW + kx];
        };
        fetch1(0, 0);
  #pragma unroll
        for (int t = 0; t < T; ++t) {
          if (t + 1 < T) fetch1(t + 1, (t + 1) & 1);
          __builtin_amdgcn_sched_barrier(0);
          acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t & 1][0], bv[t & 1][0], acc[0][0], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
      } else {
      fetch(std::integral_constant<int, 0>{});
      static_for<T>([&](auto tc) __attribute__((always_inline)) {
        constexpr int t = decltype(tc)::value;
        if constexpr (t == 10) DVC_CSTAMP(k == 3, 14);
        if constexpr (t == 26) DVC_CSTAMP(k == 3, 15);
        if constexpr (t + 1 < T) {   // operands of the next tap while this one multiplies
          fetch(std::integral_constant<int, t + 1>{});
          __builtin_amdgcn_s_waitcnt(kWaitAllButNext);
        } else {
          __builtin_amdgcn_s_waitcnt(kWaitAll);
        }
        __builtin_amdgcn_sched_barrier(0);          // the MFMAs stay behind the wait that covers their operands
  #pragma unroll
        for (int i = 0; i < MT * NT; ++i) {
          const int mt = i / NT, nt = i - mt * NT;
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t & 1][mt], bv[t & 1][nt], acc[mt][nt], 0, 0, 0);
          if constexpr (SPREAD) {
            if (i == 0) {   // behind the group's first MFMA: the pipe is busy for the next 64 cycles anyway
              if constexpr (t == 0) start_next_chunk();
              if constexpr (t < CK) { if (nextf.fast) prefetch_channel(nextf, t); }
              if constexpr (SPREAD_COMMIT && t >= T - CK) {
                if (more) commit_channel(commit_addr, std::integral_constant<int, t - (T - CK)>{});
              }
            }
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      });
      }
    }
    DVC_CSTAMP(k == 2, 10);
    if (NBUF == 1) {   // single buffers: everyone must be done reading before the next chunk lands
      __syncthreads();
      if (more) dma_weights(k + 1, 0);
    }
    if (more && !SPREAD_COMMIT) commit_input(nxt);   // (two buffers: nxt was last read before the previous barrier)
    DVC_CSTAMP(k == 2, 11);
    __builtin_amdgcn_s_waitcnt(0x0070);   /* vmcnt(0) lgkmcnt(0): this wave's DMA pieces and LDS writes have landed (a form the compiler's wait-count tracking sees) */
    DVC_CSTAMP(k == 2, 12);
    __syncthreads();
  }
  }   // (ordinary K loop)

  epilogue(lds, lane, half, l32);
}

template <int KS, int S, int MT, int NT, bool HALF = false, int BF = 0, int DY = 1>
int launch(const ConvArgs& a, int batch, hipStream_t st) {
  using G = Geo<KS, S, MT, NT, HALF, BF, DY>;
  // LDS above the 64 KiB default needs an opt-in per kernel and per device; calls may come from
  // several host threads, so the done-mask is atomic (setting the attribute twice is harmless)
  static std::atomic<unsigned long long> attr_done{0};
  int device = 0;
  if (hipGetDevice(&device) != hipSuccess || device < 0 || device > 63) device = 0;
  const unsigned long long device_bit = 1ull << device;
  if (!(attr_done.load(std::memory_order_acquire) & device_bit)) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_mfma_f32_kernel<KS, S, MT, NT, HALF, BF, DY>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES);
    if (e != hipSuccess) {
      dvc::set_error("hipFuncSetAttribute(conv %d,%d,%d,%d): %s", KS, S, MT, NT, hipGetErrorString(e));
      return DVC_EHIP;
    }
    attr_done.fetch_or(device_bit, std::memory_order_release);
  }
  ConvArgs b = a;
  const int tiles_y = (a.ho + G::TH * DY - 1) / (G::TH * DY);
  b.tiles_x = (a.wo + G::TW - 1) / G::TW;
  b.sum_slots = b.tiles_x * tiles_y;
  dim3 grid(b.tiles_x * tiles_y, HALF ? 2 * ((a.cout + 63) / 64) : (a.cout + G::CB - 1) / G::CB, batch);
  b.tiles_xy = (int)grid.x;
  b.cblocks = (int)grid.y;
  b.total_items = (int)(grid.x * grid.y * grid.z);
  if constexpr (G::WSP) {
    // persistent: one workgroup per CU (the 152 KB of LDS admit no second one), a multiple of the 8 XCDs
    static std::atomic<int> cus{0};
    int ncu = cus.load(std::memory_order_relaxed);
    if (ncu == 0) {
      if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || ncu < 8) ncu = 256;
      cus.store(ncu, std::memory_order_relaxed);
    }
    int wgs = std::min(ncu & ~7, (b.total_items + 7) & ~7);
#ifdef DVC_CONV_STAMPS
    if (const char* f = getenv("DVC_CONV_WGS")) wgs = atoi(f);   // diagnostic build: fewer CUs at work
#endif
    grid = dim3(wgs, 1, 1);
  }
  hipLaunchKernelGGL((conv_mfma_f32_kernel<KS, S, MT, NT, HALF, BF, DY>), grid, dim3(G::THREADS), G::LDS_BYTES, st, b);
  return dvc::check_launch("conv_mfma_f32_kernel");
}

// Rows of 32 pixels per wave: the largest of max_nt / ... / 1 that still yields >= 2 workgroups per CU.
inline int pick_nt(const ConvArgs& a, int batch, int mt, int max_nt) {
  const long long cblocks = (a.cout + 32 * mt - 1) / (32 * mt);
  const long long tx = (a.wo + 31) / 32;
#ifdef DVC_CONV_STAMPS
  if (const char* f = getenv("DVC_CONV_FORCE_NT")) return atoi(f);   // diagnostic build: e.g. one workgroup per CU
#endif
  for (int nt = max_nt; nt > 1; nt >>= 1) {
    const long long blocks = (long long)((a.ho + 4 * nt - 1) / (4 * nt)) * tx * cblocks * batch;
    if (blocks >= 480) return nt;   // (within a few percent of two workgroups per CU counts)
  }
  return 1;
}

// Partial sums per output plane that a 3x3 stride-1 fp32 launch can emit with its stores (ConvArgs::plane_sums):
// one per workgroup tile, when the launch is made of whole tiles of the variant dispatch_tile<3, 1, 0> picks, has
// 16-byte rows, no residual and no PixelShuffle.  0: not available for this launch.
// The tile height (and with it whether sums are available, how many slots there are and in which order the
// pixels of a plane are added up) is chosen from ONE sample's tile count, whatever the batch: the sums of a
// sample -- and the gate, and everything downstream of it -- are then the same bits whether the sample is
// coded alone or inside a batch (a stream coded as a batch must decode per sample).
inline int conv_plane_sum_slots(const ConvArgs& a, int ksize, int stride, int precision, int /*batch*/) {
  if (precision != 0 || ksize != 3 || stride != 1 || a.ps || !a.vec4 || a.res != nullptr || a.out_slope < 0.f ||
      a.out_slope > 1.f || a.cout <= 4)
    return 0;
  const int mt = a.cout > 32 ? 2 : 1;
  const int nt = pick_nt(a, 1, mt, 4);
  if (a.cout % (32 * mt) != 0 || a.ho % (4 * nt) != 0 || a.wo % 32 != 0) return 0;
  const long long slots = (long long)(a.ho / (4 * nt)) * (a.wo / 32);
  return slots <= 65536 ? (int)slots : 0;
}

template <int KS, int S, int BF = 0>
int dispatch_tile(const ConvArgs& a, int batch, hipStream_t st) {
  const int mt = a.cout > 32 ? 2 : 1;
  // stride 2 stages a (2*TH+1) x 65 halo per channel: 16-row tiles would not leave room for two
  // workgroups per CU, so those stop at 8 rows
  // (the split-operand 3x3 kernels hold two parts of every operand: 64 channels x 16 rows would spill)
  // (except stride 1, where the 16-row tile is the wave-specialised form)
  constexpr bool WIDE_OK = !(S == 2 && KS != 1) && !(BF >= 2 && KS == 3 && S != 1);
  // (a launch that emits plane sums picks its tile from one sample's tile count: see conv_plane_sum_slots)
  const int nt = pick_nt(a, a.plane_sums != nullptr ? 1 : batch, mt, S == 2 && KS != 1 ? 2 : (mt == 2 && !WIDE_OK) ? 2 : 4);
  if (mt == 2) {
    if (nt == 4) { if constexpr (WIDE_OK) return launch<KS, S, 2, 4, false, BF>(a, batch, st); }
    if (nt == 2) return launch<KS, S, 2, 2, false, BF>(a, batch, st);
    // still fewer than two workgroups per CU with the smallest tile: split the 64-channel blocks
    const long long blocks = (long long)((a.ho + 3) / 4) * ((a.wo + 31) / 32) * ((a.cout + 63) / 64) * batch;
    if constexpr (!(KS == 7 && BF != 0)) {
      if (blocks < 384 && a.plane_sums == nullptr) return launch<KS, S, 1, 1, true, BF>(a, batch, st);
    }
    return launch<KS, S, 2, 1, false, BF>(a, batch, st);
  }
  if (nt == 4) { if constexpr (!(S == 2 && KS != 1)) return launch<KS, S, 1, 4, false, BF>(a, batch, st); }
  if (nt == 2) return launch<KS, S, 1, 2, false, BF>(a, batch, st);
  return launch<KS, S, 1, 1, false, BF>(a, batch, st);
}


}  // namespace
