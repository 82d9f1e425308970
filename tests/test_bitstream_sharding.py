"""Host logic either side of the coding path: the bitstream container (dmc/test.py:38-56) and
GOP sharding over ranks (world_size 2 on gloo, CPU).  The codec inside the sharding test is the
CPU oracle standing in for the device model -- these tests check bookkeeping, not kernels."""
import io
import os
import struct

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from deepvideocodec_amd import bitstream, evaluate, sharding


def test_write_body_layout_matches_reference_framing():
    fd = io.BytesIO()
    strings = [[b"abc"], [b""], [b"\x00\x01\x02\x03\x04"]]
    n = bitstream.write_body(fd, (17, 30), strings)
    raw = fd.getvalue()
    # (shape0, shape1, n_strings) then per string (len, bytes), all big-endian uint32 framing
    expect = struct.pack(">3I", 17, 30, 3) + struct.pack(">I", 3) + b"abc" + struct.pack(">I", 0) + \
        struct.pack(">I", 5) + b"\x00\x01\x02\x03\x04"
    assert raw == expect and n == len(expect)
    shape, back = bitstream.read_body(io.BytesIO(raw))
    assert shape == (17, 30) and back == strings


def test_p_frame_round_trip_and_truncation():
    enc = {"strings": {"motion": [[b"m0"], [b"m1"], [b"mz"]], "frame": [[b"f" * 300], [b""], [b"z"]]},
           "shape": {"motion": torch.Size([1, 2]), "frame": torch.Size([3, 4])}}
    fd = io.BytesIO()
    n = bitstream.write_p_frame(fd, enc)
    assert n == len(fd.getvalue()) == 2 * 12 + 6 * 4 + 2 + 2 + 2 + 300 + 0 + 1
    strings, shapes = bitstream.read_p_frame(io.BytesIO(fd.getvalue()))
    assert strings == enc["strings"] and shapes == {"motion": (1, 2), "frame": (3, 4)}
    with pytest.raises(EOFError):
        bitstream.read_p_frame(io.BytesIO(fd.getvalue()[:-1]))


def test_gop_ranges_and_rank_assignment():
    assert sharding.gop_ranges(10, 4) == [(0, 4), (4, 8), (8, 10)]
    assert sharding.gop_ranges(0, 4) == []
    for world in (1, 2, 3, 8):
        got = sorted(g for r in range(world) for g in sharding.gops_of_rank(11, r, world))
        assert got == list(range(11))
    assert sharding.gops_of_rank(3, 5, 8) == []          # more ranks than GOPs: idle rank
    with pytest.raises(ValueError):
        sharding.gops_of_rank(3, 2, 2)
    assert sharding.order_segments([(2, b"c"), (0, b"a"), (1, b"b")]) == b"abc"


def test_pad_crop_centre():
    x = torch.arange(2 * 3 * 70 * 100, dtype=torch.float32).reshape(2, 3, 70, 100)
    p = evaluate.pad(x)
    assert p.shape == (2, 3, 128, 128)
    assert torch.equal(evaluate.crop(p, (70, 100)), x)
    assert p[:, :, :29].abs().sum() == 0 and p[:, :, :, :14].abs().sum() == 0   # (128-70)//2, (128-100)//2


class _OracleNet:
    """The CPU oracle behind DMC's encode_inter/decode_inter signature (test stand-in only)."""

    def __init__(self):
        from deepvideocodec_amd.testing import seeded_state_dict
        from oracle import dmc_oracle as O
        self.O, self.sd = O, seeded_state_dict(seed=0)
        self.ent = O.EntropyState(self.sd)
        self.ent.update(force=True)

    def encode_inter(self, x, dpb):
        return self.O.encode_inter(self.sd, self.ent, x, dpb)

    def decode_inter(self, strings, shapes, dpb):
        return self.O.decode_inter(self.sd, self.ent, strings, shapes, dpb)


def _frames():
    from deepvideocodec_amd.testing import synthetic_gop
    return synthetic_gop(6, 64, 64, seed=3)


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        stats, segs = evaluate.evaluate_sequence(_OracleNet(), _frames(), gop=2, name="s")
        q.put((rank, dict(stats), segs))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_sharded_evaluation_equals_single_process():
    """3 GOPs of 2 frames over 2 ranks: same statistics and the same bitstream as one process."""
    torch.set_num_threads(4)
    ref_stats, ref_segs = evaluate.evaluate_sequence(_OracleNet(), _frames(), gop=2, name="s")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=500) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    segs = [s for _, _, ss in got for s in ss]
    assert sorted(g for g, _ in segs) == [0, 1, 2]
    by_rank = {r: [g for g, _ in ss] for r, _, ss in got}
    assert by_rank == {0: [0, 2], 1: [1]}
    assert sharding.order_segments(segs) == sharding.order_segments(ref_segs)
    for _, stats, _ in got:   # every rank holds the all-rank statistics
        assert stats["bpp"] == pytest.approx(ref_stats["bpp"], rel=1e-12)
        # (finite: the uncoded intra pictures are not counted; the ranks run the oracle with another thread
        # count than the single process, so the fp32 reconstructions differ in the last bits)
        assert stats["psnr"] == pytest.approx(ref_stats["psnr"], rel=1e-6) and stats["psnr"] < 100
    assert ref_stats["bpp"] > 0 and list(ref_stats.keys()) == ["video", "quality", "bpp", "psnr"]
