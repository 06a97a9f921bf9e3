"""BASELINE config[0] end to end: the reference's demo (demo/fit_identity.ipynb) as a short training run on the GPU --
RSLM + LM + AMIS forward, MC loss + derivative regularisation, backward through the HIP kernels, Adam."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu


def test_fit_identity_training_reduces_pose_error():
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'demo'))
    import install as emu
    emu.uninstall()
    import fit_identity
    before, after = fit_identity.train(iters=250, batch=256, verbose=False)
    assert after[0] < 0.5 * before[0], (before, after)        # translation error halves within 250 steps
    assert after[1] < 0.7 * before[1], (before, after)
    assert after[0] < 1.5
