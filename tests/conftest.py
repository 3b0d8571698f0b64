import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle_lib import Oracle
    return Oracle.get()


@pytest.fixture(scope="session")
def ref():
    from oracle_lib import Ref
    if not Ref.available():
        pytest.skip("oracle/_ref/libtpt_ref.so not built (needs /root/reference)")
    return Ref.get()


@pytest.fixture(scope="session")
def tpt():
    """The product, initialised on the GPU (gpu tests only). Fails loudly if the HIP library is missing."""
    from toypathtracer_amd import api
    api.InitializeTest()
    yield api
    api.ShutdownTest()


@pytest.fixture()
def tpt_defaults(tpt):
    """Reset every run-time knob to its default before a test."""
    tpt.set_scene(None)
    tpt.set_camera(None)
    tpt.set_samples_per_pixel(4)
    tpt.set_seed_mode(tpt.SEED_PER_PIXEL)
    tpt.set_fold_mode(tpt.FOLD_RECURSIVE)
    tpt.set_kernel_variant(0, 3, -1)
    tpt.set_row_shard(0, 1, 0)
    tpt.set_frame_overlap(16)
    return tpt
