import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import binding
    binding.build()
    return binding


@pytest.fixture(scope="session")
def tables():
    """Seeded synthetic tables in MERL layout, built once per session."""
    from mitsuba_customization_amd import synth
    cache = {}

    def get(kind, seed=0, dims=synth.MERL_DIMS):
        key = (kind, seed, tuple(dims))
        if key not in cache:
            cache[key] = synth.make_table(kind, seed, dims)
        return cache[key]

    return get
