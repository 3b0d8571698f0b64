"""Context life cycle.  Runs last (file name): it tears the device context down and brings it back."""
import pytest

pytestmark = pytest.mark.gpu


def test_shutdown_and_reinitialise(tpt_defaults, oracle):
    """ShutdownTest releases every device resource and InitializeTest brings the context back (the reference's hosts call
    the pair once, Test.cpp:240-253; a library may see it more often)."""
    import numpy as np
    from common import oracle_frames
    from oracle_lib import SEED_PER_PIXEL
    tpt = tpt_defaults
    w, h = 96, 64
    ro, bo, _ = oracle_frames(oracle, w, h, 4, 2, seed_mode=SEED_PER_PIXEL)
    for cycle in range(3):
        bb = np.zeros((h, w, 4), np.float32)
        rays = 0
        for f in range(2):
            tpt.UpdateTest(0.0, f, w, h, 2)
            rays += tpt.DrawTest(0.0, f, w, h, bb, 2)
        assert rays == ro, cycle
        assert bb.tobytes() == bo.tobytes(), cycle
        tpt.ShutdownTest()
        tpt.InitializeTest()
