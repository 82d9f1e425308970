"""Run-to-run determinism under concurrent GPU load.  The decoder re-derives the encoder's entropy parameters,
so every stage of the coding path has to give bit-identical results when it is run again on the same inputs --
also while another host thread keeps the GPU busy from its own stream, as the benchmark's GOPs in flight do
(a kernel that passes alone can still fail here: misaligned 8-byte gathers in the flow warp once did)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*extra):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "determinism_stress.py"), *extra],
                       capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    return r.stdout


@pytest.mark.timeout(300)
@pytest.mark.parametrize("precision", ["fp32", "fp16x2"])
def test_flow_warp_repeats_identically_under_load(precision):
    """(the load thread's kernels differ with the operand mode: the persistent split-operand convolutions keep every
    CU's memory queue full, which is what exposed a store form of the warp that was only correct alone)"""
    out = _run("--precision", precision, "--warp-reps", "60")
    lines = [ln for ln in out.splitlines() if "repeats differ" in ln]
    assert len(lines) == 4, out[-2000:]
    for ln in lines:
        assert ": 0 of 60 repeats differ" in ln, out[-3000:]


@pytest.mark.timeout(300)
@pytest.mark.parametrize("precision", ["fp32", "fp16x2"])
def test_coding_stages_repeat_identically_under_load(precision):
    out = _run("--precision", precision, "--frames", "3")
    assert "DIFFERENT stages: none" in out, out[-3000:]
