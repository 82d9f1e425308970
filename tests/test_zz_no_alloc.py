"""dvc_se_gate_static: the squeeze-excite gate with the library's built-in scratch (no allocation inside
the call) against dvc_se_gate with caller scratch -- same kernels, so the gates must be bit-identical.
Two calls in a row on one stream, the second with a smaller input (the shared scratch keeps the first
call's partial sums), then the size limit."""
import ctypes as C

import pytest
import torch

from deepvideocodec_amd import _lib, ops

pytestmark = pytest.mark.gpu


def _gate_both(x, fc1, fc2):
    lib = _lib.load()
    n, c, h, w = x.shape
    stream = torch.cuda.current_stream().cuda_stream
    want = ops.se_gate(x, fc1, fc2)                                 # caller-provided scratch
    got = torch.full((n * c,), float("nan"), dtype=torch.float32, device=x.device)   # pre-allocated by the caller
    src = ops._src(x)
    rc = lib.dvc_se_gate_static(C.byref(src), n, h, w, fc1.data_ptr(), fc2.data_ptr(), fc1.size(0),
                                got.data_ptr(), stream)
    assert rc == 0, _lib.last_error()
    return want, got


def test_static_scratch_gate_equals_caller_scratch_gate():
    g = torch.Generator().manual_seed(5)
    dev = torch.device("cuda", 0)
    for (n, c, h, w) in ((2, 128, 68, 120), (1, 32, 17, 31)):       # second call: fewer planes, ragged plane size
        x = torch.randn(n, c, h, w, generator=g).to(dev)
        fc1 = (torch.randn(c // 16, c, generator=g) * 0.3).to(dev)
        fc2 = (torch.randn(c, c // 16, generator=g) * 0.3).to(dev)
        want, got = _gate_both(x, fc1, fc2)
        torch.cuda.synchronize()
        assert torch.equal(got, want), (n, c, h, w)
        ref = torch.sigmoid(torch.relu(x.double().mean(dim=(2, 3)) @ fc1.double().t()) @ fc2.double().t()).reshape(-1)
        assert (got.double() - ref).abs().max().item() < 1e-6


def test_static_scratch_gate_rejects_oversize():
    lib = _lib.load()
    dev = torch.device("cuda", 0)
    n, c = 17, 256                                                 # 4352 planes > 4096
    x = torch.zeros(n, c, 2, 2, device=dev)
    fc1 = torch.zeros(16, c, device=dev)
    fc2 = torch.zeros(c, 16, device=dev)
    gate = torch.zeros(n * c, device=dev)
    rc = lib.dvc_se_gate_static(C.byref(ops._src(x)), n, 2, 2, fc1.data_ptr(), fc2.data_ptr(), 16, gate.data_ptr(),
                                torch.cuda.current_stream().cuda_stream)
    assert rc == -1 and "exceeds the built-in scratch" in _lib.last_error()


def test_static_scratch_gate_allocates_nothing_and_captures():
    """What the entry point exists for: a (warm) call leaves the device's free memory and torch's allocation counter
    where they were, and it can be recorded into a HIP graph (an allocation inside capture would fail it) whose
    replay gives the same bits.  The first call on a device looks the scratch address up and must be made outside
    capture; this test makes it first."""
    lib = _lib.load()
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(8)
    n, c, h, w = 1, 64, 68, 120
    x = torch.randn(n, c, h, w, generator=g).to(dev)
    fc1 = (torch.randn(c // 16, c, generator=g) * 0.3).to(dev)
    fc2 = (torch.randn(c, c // 16, generator=g) * 0.3).to(dev)
    want = ops.se_gate(x, fc1, fc2)
    got = torch.zeros(n * c, device=dev)
    src = ops._src(x)

    def call(stream):
        rc = lib.dvc_se_gate_static(C.byref(src), n, h, w, fc1.data_ptr(), fc2.data_ptr(), fc1.size(0), got.data_ptr(), stream)
        assert rc == 0, _lib.last_error()

    call(torch.cuda.current_stream().cuda_stream)          # warm: per-device symbol lookup happens here
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info(dev)
    count0 = torch.cuda.memory_stats(dev)["allocation.all.allocated"]
    got.zero_()
    call(torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info(dev)
    assert free1 == free0, (free0, free1)
    assert torch.cuda.memory_stats(dev)["allocation.all.allocated"] == count0
    assert torch.equal(got, want)
    # under stream capture
    got.zero_()
    graph = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        with torch.cuda.graph(graph, stream=side):
            call(side.cuda_stream)
    torch.cuda.current_stream().wait_stream(side)
    assert float(got.abs().max()) == 0.0                   # capture records, it does not run
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(got, want)
