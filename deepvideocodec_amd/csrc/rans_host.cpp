// Host side of libdvc_hip.so: error reporting, device introspection and the range
// coder (64-bit-state rANS, 32-bit little-endian words, 16-bit CDF precision,
// 4-bit bypass escape for symbols outside the tabulated support).
//
// Stream layout: the encoder walks the symbol records backwards and emits words
// from the end of its buffer towards the front, finishing with the 64-bit state
// (low word first); the decoder therefore reads strictly forwards.
#include <hip/hip_runtime_api.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <vector>

#include "dvc_common.h"

namespace {
thread_local char g_err[512] = "";

constexpr int kPrecision = 16;
constexpr int kBypassBits = 4;
constexpr uint32_t kMaxBypass = (1u << kBypassBits) - 1;
constexpr uint64_t kRansL = 1ull << 31;


inline void put(uint64_t& x, uint32_t*& ptr, uint32_t start, uint32_t freq) {
  const uint64_t x_max = ((kRansL >> kPrecision) << 32) * freq;
  if (x >= x_max) {
    *--ptr = (uint32_t)x;
    x >>= 32;
  }
  x = ((x / freq) << kPrecision) + (x % freq) + start;
}

inline void put_bits(uint64_t& x, uint32_t*& ptr, uint32_t val) {
  const uint32_t freq = 1u << (16 - kBypassBits);
  const uint64_t x_max = ((kRansL >> 16) << 32) * freq;
  if (x >= x_max) {
    *--ptr = (uint32_t)x;
    x >>= 32;
  }
  x = (x << kBypassBits) | val;
}

struct Reader {
  const uint32_t* p;
  const uint32_t* end;
  uint64_t x;
  bool ok = true;
  inline void renorm() {
    if (x < kRansL) {
      if (p >= end) { ok = false; return; }
      x = (x << 32) | *p++;
    }
  }
  inline uint32_t bits() {
    const uint32_t v = (uint32_t)(x & ((1u << kBypassBits) - 1));
    x >>= kBypassBits;
    renorm();
    return v;
  }
};
}  // namespace

namespace dvc {
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace dvc

extern "C" const char* dvc_last_error(void) { return g_err; }
extern "C" const char* dvc_version(void) { return "dvc_hip 0.1 (gfx950)"; }

extern "C" int dvc_device_count(void) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) {
    dvc::set_error("hipGetDeviceCount: %s", hipGetErrorString(e));
    return DVC_EHIP;
  }
  return n;
}

extern "C" int64_t dvc_rans_encode(const int32_t* symbols, const int32_t* indexes, int64_t n, const int32_t* cdfs,
                                   int32_t n_cdfs, int32_t cdf_stride, const int32_t* cdf_sizes,
                                   const int32_t* offsets, uint8_t* out, int64_t out_capacity) {
  DVC_REQUIRE(n >= 0 && (n == 0 || (symbols && indexes)) && cdfs && cdf_sizes && offsets && out,
              "dvc_rans_encode: null pointer");
  // Validate the tables once (the per-symbol loop then trusts them).
  for (int32_t c = 0; c < n_cdfs; ++c) {
    const int32_t* cdf = cdfs + (int64_t)c * cdf_stride;
    const int32_t size = cdf_sizes[c];
    DVC_REQUIRE(size >= 2 && size <= cdf_stride, "dvc_rans_encode: bad cdf size %d for table %d", size, c);
    for (int32_t s = 0; s + 1 < size; ++s)
      DVC_REQUIRE(cdf[s + 1] > cdf[s] && cdf[s] >= 0 && cdf[s + 1] <= (1 << kPrecision),
                  "dvc_rans_encode: table %d is not strictly increasing within [0, 2^16]", c);
  }
  DVC_REQUIRE(out_capacity >= 8 && reinterpret_cast<uintptr_t>(out) % 4 == 0,
              "dvc_rans_encode: output buffer must be 4-byte aligned and hold >= 8 bytes");
  for (int64_t i = 0; i < n; ++i)
    DVC_REQUIRE(indexes[i] >= 0 && indexes[i] < n_cdfs, "dvc_rans_encode: index %d out of range at %lld", indexes[i], (long long)i);
  // One backward pass, words written from the end of `out` towards its start.  A symbol's
  // records go out in reverse of their decode order: raw nibbles (last first), the nibble
  // count, then the table symbol itself.
  uint32_t* const buf_begin = reinterpret_cast<uint32_t*>(out);
  uint32_t* const buf_end = buf_begin + out_capacity / 4;
  uint32_t* ptr = buf_end;
  uint64_t x = kRansL;
  for (int64_t i = n; i-- > 0;) {
    if (ptr - buf_begin < 16) {   // worst case per symbol: 1 + 10 nibble records, each <= 1 word
      dvc::set_error("dvc_rans_encode: output capacity %lld bytes exhausted", (long long)out_capacity);
      return DVC_ENOSPC;
    }
    const int32_t idx = indexes[i];
    const int32_t* cdf = cdfs + (int64_t)idx * cdf_stride;
    const int32_t max_value = cdf_sizes[idx] - 2;
    int32_t value = symbols[i] - offsets[idx];
    if (value < 0 || value >= max_value) {
      uint32_t raw;
      if (value < 0) raw = (uint32_t)(-2 * (int64_t)value - 1);
      else raw = (uint32_t)(2 * ((int64_t)value - max_value));
      value = max_value;
      int32_t n_bypass = 0;
      while (n_bypass < 8 && (raw >> (n_bypass * kBypassBits)) != 0) ++n_bypass;
      for (int32_t j = n_bypass; j-- > 0;) put_bits(x, ptr, (raw >> (j * kBypassBits)) & kMaxBypass);
      // the count is sent as (n_bypass / 15) nibbles of 15 followed by n_bypass % 15
      put_bits(x, ptr, (uint32_t)n_bypass % kMaxBypass);
      for (int32_t q = n_bypass / (int32_t)kMaxBypass; q-- > 0;) put_bits(x, ptr, kMaxBypass);
    }
    put(x, ptr, (uint32_t)cdf[value], (uint32_t)(cdf[value + 1] - cdf[value]));
  }
  ptr -= 2;
  ptr[0] = (uint32_t)x;
  ptr[1] = (uint32_t)(x >> 32);
  const int64_t nbytes = (int64_t)(buf_end - ptr) * 4;
  memmove(out, ptr, (size_t)nbytes);
  return nbytes;
}

extern "C" int dvc_rans_decode(const uint8_t* data, int64_t n_bytes, const int32_t* indexes, int64_t n,
                               const int32_t* cdfs, int32_t n_cdfs, int32_t cdf_stride, const int32_t* cdf_sizes,
                               const int32_t* offsets, int32_t* symbols_out) {
  DVC_REQUIRE(data && cdfs && cdf_sizes && offsets && (n == 0 || (indexes && symbols_out)),
              "dvc_rans_decode: null pointer");
  if (n_bytes < 8 || n_bytes % 4 != 0) {
    dvc::set_error("dvc_rans_decode: stream of %lld bytes is not a whole number (>= 2) of 32-bit words", (long long)n_bytes);
    return DVC_EFORMAT;
  }
  std::vector<uint32_t> words((size_t)n_bytes / 4);
  memcpy(words.data(), data, (size_t)n_bytes);
  Reader rd{words.data() + 2, words.data() + words.size(), (uint64_t)words[0] | ((uint64_t)words[1] << 32)};
  const uint32_t mask = (1u << kPrecision) - 1;
  // Validate the tables once, and build per table a 257-entry bucket index: first[b] = largest s
  // with cdf[s] <= b * 256.  The symbol of a cumulative value cum then lies in
  // [first[cum >> 8], first[(cum >> 8) + 1]], a range of one or two entries for peaked tables.
  constexpr int kBuckets = 256;
  std::vector<uint16_t> first((size_t)n_cdfs * (kBuckets + 1));
  for (int32_t c = 0; c < n_cdfs; ++c) {
    const int32_t* cdf = cdfs + (int64_t)c * cdf_stride;
    const int32_t size = cdf_sizes[c];
    DVC_REQUIRE(size >= 2 && size <= cdf_stride && size <= 65536, "dvc_rans_decode: bad cdf size %d for table %d", size, c);
    DVC_REQUIRE(cdf[0] == 0 && cdf[size - 1] == (1 << kPrecision), "dvc_rans_decode: table %d does not span [0, 2^16]", c);
    int32_t s = 0;
    uint16_t* f = first.data() + (size_t)c * (kBuckets + 1);
    for (int b = 0; b <= kBuckets; ++b) {
      const int32_t v = b << 8;
      while (s + 1 < size - 1 && cdf[s + 1] <= v) {
        DVC_REQUIRE(cdf[s + 1] >= cdf[s], "dvc_rans_decode: table %d is not monotone", c);
        ++s;
      }
      f[b] = (uint16_t)s;
    }
  }
  for (int64_t i = 0; i < n; ++i)
    DVC_REQUIRE(indexes[i] >= 0 && indexes[i] < n_cdfs, "dvc_rans_decode: index %d out of range at %lld", indexes[i], (long long)i);
  // the zero symbol of every table (position -offset): one range check decodes it without a table search
  struct Zero { uint32_t start, freq; int32_t value; };
  std::vector<Zero> zero((size_t)n_cdfs);
  for (int32_t c = 0; c < n_cdfs; ++c) {
    const int32_t* cdf = cdfs + (int64_t)c * cdf_stride;
    const int64_t pos = -(int64_t)offsets[c];
    zero[c] = Zero{0, 0, 0};
    if (pos >= 0 && pos < cdf_sizes[c] - 2) zero[c] = Zero{(uint32_t)cdf[pos], (uint32_t)(cdf[pos + 1] - cdf[pos]), 0};
  }
  for (int64_t i = 0; i < n; ++i) {
    const int32_t idx = indexes[i];
    const uint32_t cum = (uint32_t)(rd.x & mask);
    {
      const Zero& z = zero[idx];
      if (cum - z.start < z.freq) {     // (unsigned: also false for cum < start; freq 0 disables the entry)
        rd.x = (uint64_t)z.freq * (rd.x >> kPrecision) + cum - z.start;
        rd.renorm();
        if (!rd.ok) {
          dvc::set_error("dvc_rans_decode: bitstream exhausted or malformed at symbol %lld", (long long)i);
          return DVC_EFORMAT;
        }
        symbols_out[i] = 0;
        continue;
      }
    }
    const int32_t* cdf = cdfs + (int64_t)idx * cdf_stride;
    const int32_t max_value = cdf_sizes[idx] - 2;
    // largest s with cdf[s] <= cum: bucket bounds, then bisection inside the bucket
    const uint16_t* f = first.data() + (size_t)idx * (kBuckets + 1) + (cum >> 8);
    int32_t lo = f[0], hi = f[1] + 1;
    while (hi - lo > 1) {
      const int32_t mid = (lo + hi) >> 1;
      if ((uint32_t)cdf[mid] <= cum) lo = mid; else hi = mid;
    }
    const int32_t s = lo;
    const uint32_t start = (uint32_t)cdf[s], freq = (uint32_t)(cdf[s + 1] - cdf[s]);
    rd.x = (uint64_t)freq * (rd.x >> kPrecision) + (rd.x & mask) - start;
    rd.renorm();
    int32_t value = s;
    if (value == max_value) {
      uint32_t val = rd.bits();
      int32_t n_bypass = (int32_t)val;
      while (val == kMaxBypass && rd.ok) {
        val = rd.bits();
        n_bypass += (int32_t)val;
      }
      if (n_bypass > 8) rd.ok = false;
      uint32_t raw = 0;
      for (int32_t j = 0; j < n_bypass && rd.ok; ++j) raw |= rd.bits() << (j * kBypassBits);
      value = (int32_t)(raw >> 1);
      if (raw & 1) value = -value - 1; else value += max_value;
    }
    if (!rd.ok) {
      dvc::set_error("dvc_rans_decode: bitstream exhausted or malformed at symbol %lld", (long long)i);
      return DVC_EFORMAT;
    }
    symbols_out[i] = value + offsets[idx];
  }
  return DVC_OK;
}

extern "C" int dvc_pmf_to_quantized_cdf(const float* pmf, int32_t n_pmf, int32_t precision, int32_t* cdf_out) {
  DVC_REQUIRE(pmf && cdf_out && n_pmf >= 1 && precision >= 1 && precision <= 16, "dvc_pmf_to_quantized_cdf: bad args");
  DVC_REQUIRE(n_pmf + 1 <= (1 << precision), "dvc_pmf_to_quantized_cdf: more symbols than 2^precision counts");
  std::vector<uint32_t> cdf((size_t)n_pmf + 1);
  cdf[0] = 0;
  uint64_t total = 0;
  for (int32_t i = 0; i < n_pmf; ++i) {
    const double v = (double)pmf[i] * (double)(1 << precision);
    cdf[i + 1] = (uint32_t)(v + 0.5);  // round half away from zero, pmf >= 0
    total += cdf[i + 1];
  }
  DVC_REQUIRE(total != 0, "dvc_pmf_to_quantized_cdf: pmf sums to zero");
  for (auto& v : cdf) v = (uint32_t)((((uint64_t)1 << precision) * v) / total);
  for (size_t i = 1; i < cdf.size(); ++i) cdf[i] += cdf[i - 1];
  cdf.back() = 1u << precision;
  for (int32_t i = 0; i < n_pmf; ++i) {
    if (cdf[i] == cdf[i + 1]) {
      uint32_t best_freq = ~0u;
      int32_t best = -1;
      for (int32_t j = 0; j < n_pmf; ++j) {
        const uint32_t freq = cdf[j + 1] - cdf[j];
        if (freq > 1 && freq < best_freq) { best_freq = freq; best = j; }
      }
      DVC_REQUIRE(best != -1, "dvc_pmf_to_quantized_cdf: cannot give every symbol a non-zero count");
      if (best < i) for (int32_t j = best + 1; j <= i; ++j) cdf[j]--;
      else for (int32_t j = i + 1; j <= best; ++j) cdf[j]++;
    }
  }
  for (size_t i = 0; i < cdf.size(); ++i) cdf_out[i] = (int32_t)cdf[i];
  return DVC_OK;
}
