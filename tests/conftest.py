import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run on the GPU box with -m gpu)')


def pytest_collection_modifyitems(config, items):
    """A hung kernel or rendezvous must fail ONE test, not eat the whole run: every test gets a generous wall-clock limit
    (pytest-timeout; the slowest test -- two SFT processes on one GPU -- takes about 80 s)."""
    if config.pluginmanager.hasplugin('timeout'):
        for it in items:
            if it.get_closest_marker('timeout') is None:
                it.add_marker(pytest.mark.timeout(900))


@pytest.fixture(scope='session')
def golden_dir():
    return GOLDEN


@pytest.fixture(scope='session')
def golden_model():
    """Truncated true-width Vlaser-2B (+VLA head) with the deterministic synthetic weights of the golden fixtures."""
    import torch
    from vlaser_amd import config as C, synth
    torch.set_grad_enabled(False)
    cfg = C.truncated(C.vlaser_2b(), 2, 2)
    vla = C.VLAConfig(base=cfg)
    sd = synth.vla_state_dict(vla, with_head=True)
    return cfg, vla, sd
