import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    with np.load(os.path.join(GOLDEN, name)) as z:
        return {k: z[k] for k in z.files}


@pytest.fixture(scope="session")
def seeded_sd():
    from deepvideocodec_amd.testing import seeded_state_dict
    return seeded_state_dict(seed=0)


@pytest.fixture(scope="session")
def oracle_entropy(seeded_sd):
    from oracle.dmc_oracle import EntropyState
    ent = EntropyState(seeded_sd)
    ent.update(force=True)
    return ent


@pytest.fixture(scope="session")
def gpu_net(seeded_sd):
    """The product DMC on cuda:0 with the seeded weights and CDF tables built."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from deepvideocodec_amd.models import DMC
    net = DMC()
    net.load_state_dict(seeded_sd)
    net = net.to("cuda:0").eval()
    net.update(force=True)
    import os
    net.set_precision(os.environ.get("DVC_TEST_PRECISION", "fp32"))
    return net
