"""Training parity on the MI355X: the Trainer's loss / PSNR curves over identical steps against the CPU oracle's
autograd + torch.optim.Adam loop (= the reference's Runner.train arithmetic)."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "probes"))


@pytest.mark.parametrize("precision,loss_tol,psnr_tol", [("fp32", 1e-3, 0.02), ("bf16", 3e-2, 0.3)])
def test_training_curves_follow_the_oracle(precision, loss_tol, psnr_tol):
    import train_parity
    ref, got = train_parity.run(steps=24, B=32, precision=precision)
    assert np.isfinite(got).all()
    rel = np.abs(got[:, 0] - ref[:, 0]) / np.abs(ref[:, 0])
    assert rel.max() < loss_tol, (precision, rel.max(), rel.argmax())
    assert np.abs(got[:, 1] - ref[:, 1]).max() < psnr_tol, (precision, np.abs(got[:, 1] - ref[:, 1]).max())
    assert ref[-1, 0] < ref[0, 0]                      # and the loss is actually going down
