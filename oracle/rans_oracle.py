"""Pure-Python restatement of the range-ANS coder the reference reaches through
`compressai.ans` (`RansEncoder.encode_with_indexes`,
`RansDecoder.decode_with_indexes`), used by
dmc/models/video_model.py:238-239,250-251,257,273,283.

TEST INFRASTRUCTURE ONLY.  **Parity unpinned**: `compressai` (C++/pybind11,
un-vendored, un-pinned) is absent from the build environment, so byte
compatibility with it cannot be checked; the reference holds no bitstream
fixture.  This restates the published algorithm: ryg's 64-bit-state rANS
emitting 32-bit little-endian words, 16-bit CDF precision, and a 4-bit
"bypass" escape for symbols outside the tabulated CDF support.

Only suitable for small inputs (interpreted Python, one symbol at a time).
"""
import struct
from typing import List, Sequence

PRECISION = 16
BYPASS_PRECISION = 4
MAX_BYPASS_VAL = (1 << BYPASS_PRECISION) - 1
RANS64_L = 1 << 31
_M64 = (1 << 64) - 1


def _symbol_records(symbols, indexes, cdfs, cdf_sizes, offsets):
    """(start, range, is_bypass) records in forward order."""
    recs = []
    for sym, idx in zip(symbols, indexes):
        cdf = cdfs[idx]
        max_value = cdf_sizes[idx] - 2
        value = sym - offsets[idx]
        raw_val = 0
        if value < 0:
            raw_val = -2 * value - 1
            value = max_value
        elif value >= max_value:
            raw_val = 2 * (value - max_value)
            value = max_value
        recs.append((cdf[value], cdf[value + 1] - cdf[value], False))
        if value == max_value:
            n_bypass = 0
            while (raw_val >> (n_bypass * BYPASS_PRECISION)) != 0:
                n_bypass += 1
            val = n_bypass
            while val >= MAX_BYPASS_VAL:
                recs.append((MAX_BYPASS_VAL, MAX_BYPASS_VAL + 1, True))
                val -= MAX_BYPASS_VAL
            recs.append((val, val + 1, True))
            for j in range(n_bypass):
                v = (raw_val >> (j * BYPASS_PRECISION)) & MAX_BYPASS_VAL
                recs.append((v, v + 1, True))
    return recs


def encode_with_indexes(symbols: Sequence[int], indexes: Sequence[int],
                        cdfs: Sequence[Sequence[int]], cdf_sizes: Sequence[int],
                        offsets: Sequence[int]) -> bytes:
    recs = _symbol_records(symbols, indexes, cdfs, cdf_sizes, offsets)
    x = RANS64_L
    words: List[int] = []          # emitted back to front
    for start, rng, bypass in reversed(recs):
        if not bypass:
            x_max = ((RANS64_L >> PRECISION) << 32) * rng
            if x >= x_max:
                words.append(x & 0xFFFFFFFF)
                x >>= 32
            x = ((x // rng) << PRECISION) + (x % rng) + start
        else:
            freq = 1 << (16 - BYPASS_PRECISION)
            x_max = ((RANS64_L >> 16) << 32) * freq
            if x >= x_max:
                words.append(x & 0xFFFFFFFF)
                x >>= 32
            x = ((x << BYPASS_PRECISION) | start) & _M64
    words.append((x >> 32) & 0xFFFFFFFF)
    words.append(x & 0xFFFFFFFF)
    words.reverse()
    return struct.pack("<%dI" % len(words), *words)


class _Reader:
    def __init__(self, data: bytes):
        assert len(data) % 4 == 0 and len(data) >= 8
        self.words = struct.unpack("<%dI" % (len(data) // 4), data)
        self.pos = 2
        self.x = self.words[0] | (self.words[1] << 32)

    def _renorm(self):
        if self.x < RANS64_L:
            self.x = (self.x << 32) | self.words[self.pos]
            self.pos += 1

    def get(self):
        return self.x & ((1 << PRECISION) - 1)

    def advance(self, start, freq):
        mask = (1 << PRECISION) - 1
        self.x = freq * (self.x >> PRECISION) + (self.x & mask) - start
        self._renorm()

    def get_bits(self, n):
        val = self.x & ((1 << n) - 1)
        self.x >>= n
        self._renorm()
        return val


def decode_with_indexes(data: bytes, indexes: Sequence[int],
                        cdfs: Sequence[Sequence[int]], cdf_sizes: Sequence[int],
                        offsets: Sequence[int]) -> List[int]:
    rd = _Reader(data)
    out = []
    for idx in indexes:
        cdf = cdfs[idx]
        size = cdf_sizes[idx]
        max_value = size - 2
        cum = rd.get()
        s = 0
        while s + 1 < size and cdf[s + 1] <= cum:
            s += 1
        rd.advance(cdf[s], cdf[s + 1] - cdf[s])
        value = s
        if value == max_value:
            val = rd.get_bits(BYPASS_PRECISION)
            n_bypass = val
            while val == MAX_BYPASS_VAL:
                val = rd.get_bits(BYPASS_PRECISION)
                n_bypass += val
            raw_val = 0
            for j in range(n_bypass):
                val = rd.get_bits(BYPASS_PRECISION)
                raw_val |= val << (j * BYPASS_PRECISION)
            value = raw_val >> 1
            if raw_val & 1:
                value = -value - 1
            else:
                value += max_value
        out.append(value + offsets[idx])
    return out
