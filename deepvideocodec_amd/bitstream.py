"""Bitstream container of the evaluation loop (reference: dmc/test.py:38-56, 175, 200-203).

Everything is big-endian uint32 framing around the coder's byte strings:

  body   := shape[0] shape[1] n_strings  ( len(string) string-bytes ) * n_strings
  P frame := body("motion") body("frame")          (dict order of encode_inter's output)

The reference only writes (its decoder consumes the in-memory strings); the readers here are
the inverse of its writers so that a .bits file can be decoded on its own.
"""
import struct
from typing import BinaryIO, Dict, List, Sequence, Tuple


def write_uints(fd: BinaryIO, values: Sequence[int]) -> int:
    """dmc/test.py:38-40."""
    fd.write(struct.pack(">%dI" % len(values), *values))
    return 4 * len(values)


def write_bytes(fd: BinaryIO, values: bytes) -> int:
    """dmc/test.py:43-47 (an empty string writes nothing; counted as 0 here, None there)."""
    if len(values) == 0:
        return 0
    fd.write(values)
    return len(values)


def write_body(fd: BinaryIO, shape, out_strings) -> int:
    """dmc/test.py:50-56: `out_strings` is a list of per-batch lists; only batch item 0 is stored."""
    n = write_uints(fd, (int(shape[0]), int(shape[1]), len(out_strings)))
    for s in out_strings:
        n += write_uints(fd, (len(s[0]),))
        n += write_bytes(fd, s[0])
    return n


def write_p_frame(fd: BinaryIO, enc: Dict) -> int:
    """dmc/test.py:200-203: one body per entry of enc["shape"] / enc["strings"], in dict order."""
    n = 0
    for (_, shape), (_, strings) in zip(enc["shape"].items(), enc["strings"].items()):
        n += write_body(fd, shape, strings)
    return n


def read_uints(fd: BinaryIO, n: int) -> Tuple[int, ...]:
    raw = fd.read(4 * n)
    if len(raw) != 4 * n:
        raise EOFError("bitstream truncated inside a uint field")
    return struct.unpack(">%dI" % n, raw)


def read_body(fd: BinaryIO) -> Tuple[Tuple[int, int], List[List[bytes]]]:
    """Inverse of write_body; returns (shape, strings) with strings in encode_inter's list-of-lists form."""
    h, w, n_strings = read_uints(fd, 3)
    strings = []
    for _ in range(n_strings):
        (length,) = read_uints(fd, 1)
        s = fd.read(length)
        if len(s) != length:
            raise EOFError("bitstream truncated inside a string")
        strings.append([s])
    return (h, w), strings


def read_p_frame(fd: BinaryIO, keys=("motion", "frame")) -> Tuple[Dict, Dict]:
    """Inverse of write_p_frame: (strings, shapes) ready for DMC.decode_inter."""
    strings, shapes = {}, {}
    for k in keys:
        shapes[k], strings[k] = read_body(fd)
    return strings, shapes
