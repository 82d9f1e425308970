"""The hand-placed LDS waits of the convolution kernels, checked on the machine code (tools/check_lds_waits.py):
no instruction may touch the destination of an LDS read between its issue and the s_waitcnt that covers it, and the
fp32 kernels use no private memory.  CPU only; needs the objects a build leaves under deepvideocodec_amd/lib/obj."""
import glob
import importlib.util
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("check_lds_waits", os.path.join(ROOT, "tools", "check_lds_waits.py"))
chk = importlib.util.module_from_spec(spec)
spec.loader.exec_module(chk)

LISTING = """
0000000000001000 <kernel_a>:
\tds_read_b32 v5, v9 offset:16                               // 000000001000: D86C0010 05000009
\tds_read_b128 v[10:13], v9                                  // 000000001008: D9FE0000 0A000009
\tv_add_f32_e32 v1, v2, v3                                   // 000000001010: 02020702
\ts_waitcnt lgkmcnt(1)                                       // 000000001014: BF8CC17F
\tv_mov_b32_e32 v6, v5                                       // 000000001018: 7E0C0305
{extra}
\ts_waitcnt lgkmcnt(0)                                       // 000000001024: BF8CC07F
\tv_mfma_f32_32x32x2_f32 a[0:15], v10, v11, a[0:15]          // 000000001028: D3C00000 0402170A
\ts_endpgm                                                   // 000000001030: BF810000
"""


def run(extra):
    funcs = chk.parse(LISTING.format(extra=extra))
    assert list(funcs) == ["kernel_a"]
    return chk.check_function("kernel_a", funcs["kernel_a"])


def test_checker_accepts_covered_reads_and_flags_early_uses():
    n, problems = run("\tv_add_f32_e32 v1, v2, v3                                   // 00000000101C: 02020702")
    assert n == 2 and problems == []
    # a copy of a register of the 16-byte read before the lgkmcnt(0) that covers it
    n, problems = run("\tv_mov_b32_e32 v7, v12                                      // 00000000101C: 7E0E030C")
    assert len(problems) == 1 and "v[12]" in problems[0]
    # an overwrite counts as well
    n, problems = run("\tv_mov_b32_e32 v13, v2                                      // 00000000101C: 7E1A0302")
    assert len(problems) == 1


def test_checker_follows_loops():
    listing = """
0000000000002000 <kernel_b>:
\tv_mov_b32_e32 v1, v5                                       // 000000002000: 7E020305
\tds_read_b32 v5, v9                                         // 000000002004: D86C0000 05000009
\ts_cbranch_scc1 65533                                       // 00000000200C: BF85FFFD <kernel_b+0x0>
\ts_waitcnt lgkmcnt(0)                                       // 000000002010: BF8CC07F
\ts_endpgm                                                   // 000000002014: BF810000
"""
    funcs = chk.parse(listing)
    n, problems = chk.check_function("kernel_b", funcs["kernel_b"])
    assert len(problems) == 1, problems   # the read pending at the back edge meets the copy at the loop head


@pytest.mark.timeout(600)
def test_built_fp32_convolution_objects_hold_their_waits():
    objs = sorted(glob.glob(os.path.join(ROOT, "deepvideocodec_amd", "lib", "obj", "conv_mfma_f32_*.hip.o")))
    if not objs or not os.path.exists(os.path.join(chk.LLVM, "llvm-objdump")):
        pytest.skip("no built objects (make -C deepvideocodec_amd/csrc) or no llvm-objdump")
    import subprocess
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        for obj in objs:
            co = chk.extract_device_code(obj, tmp)
            dis = subprocess.run([os.path.join(chk.LLVM, "llvm-objdump"), "-d", co], capture_output=True, text=True,
                                 check=True).stdout
            total = 0
            for name, insns in chk.parse(dis).items():
                n, problems = chk.check_function(name, insns)
                assert problems == [], (os.path.basename(obj), problems[:5])
                total += n
            assert total > 1000, (obj, total)   # the K loops' reads were seen at all


def test_built_warp_object_holds_its_waits():
    """flow_warp_split_kernel reads its LDS windows with explicit ds_read_b32 behind one hand-placed wait."""
    obj = os.path.join(ROOT, "deepvideocodec_amd", "lib", "obj", "warp_resample.hip.o")
    if not os.path.exists(obj) or not os.path.exists(os.path.join(chk.LLVM, "llvm-objdump")):
        pytest.skip("no built object (make -C deepvideocodec_amd/csrc) or no llvm-objdump")
    import subprocess
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        co = chk.extract_device_code(obj, tmp)
        dis = subprocess.run([os.path.join(chk.LLVM, "llvm-objdump"), "-d", co], capture_output=True, text=True,
                             check=True).stdout
        seen = 0
        for name, insns in chk.parse(dis).items():
            n, problems = chk.check_function(name, insns)
            assert problems == [], (name, problems[:5])
            if "flow_warp_split_kernel" in name:
                seen = n
                assert not any(m.startswith("scratch_") for _, m, _, _ in insns), "private memory in the split warp kernel"
        assert seen >= 48, seen   # three unrolled ring slots x sixteen reads
