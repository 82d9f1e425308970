"""Block-level parity on the GPU against fixtures captured from the reference's own modules
(tests/golden/blocks_64.npz, written by make_golden.py from dmc/models/layers.py:236-366 and
dmc/models/video_model.py:37-65): SpyNet flow, motion encoder / decoder, one UNet of the reconstruction
net, multi-scale context fusion.  A whole-path failure localises to one of these."""
import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu

TOL = 1e-5


def T(a):
    return torch.from_numpy(np.asarray(a)).cuda()


def _close(got, ref, name):
    err = (got - ref).abs().max().item()
    assert err <= TOL * max(1.0, ref.abs().max().item()), f"{name}: max |diff| {err:.3e} (ref max {ref.abs().max().item():.3f})"


@pytest.fixture(autouse=True)
def _inference_mode():
    with torch.no_grad():
        yield


def test_spynet_flow(gpu_net):
    g = load_golden("blocks_64.npz")
    _close(gpu_net.optic_flow(T(g["x1"]), T(g["x0"])), T(g["spynet_mv"]), "ME_Spynet")


def test_motion_encoder_decoder(gpu_net):
    g = load_golden("blocks_64.npz")
    _close(gpu_net.motion_encoder(T(g["spynet_mv"])), T(g["motion_enc"]), "motion_encoder")
    _close(gpu_net.motion_decoder(T(g["motion_enc"])), T(g["motion_dec"]), "motion_decoder")


def test_unet(gpu_net):
    g = load_golden("blocks_64.npz")
    _close(gpu_net.recon_generation_net.unet_1(T(g["unet_in"])), T(g["unet_out"]), "UNet")


def test_context_fusion(gpu_net):
    g = load_golden("blocks_64.npz")
    out = gpu_net.context_fusion_net(T(g["fusion_in0"]), T(g["fusion_in1"]), T(g["fusion_in2"]))
    for i in range(3):
        _close(out[i], T(g[f"fusion_out{i}"]), f"MultiScaleContextFusion output {i}")
