"""GPU parity of the training step (train_IAN.py:47-352 restated in oracle/train_twin.py) through the C ABI of
include/ian_train.h.  The reference is float64 autograd on the CPU twin; tolerance on gradients is relative to the
largest entry of each tensor (fp32 chains of ~25 layers with batch-statistics batch-norm)."""
import os

import numpy as np
import pytest

from oracle import ian_oracle as O
from oracle.train_twin import TrainTwin, make_train_params

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CFG = os.path.join(ROOT, "neural_photo_editor_amd", "configs", "IAN.py")
B = 4
TOL_LOSS = 2e-4
TOL_GRAD = 3e-3


def rel(a, b):
    return float(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max() / (np.abs(np.asarray(b)).max() + 1e-30))


def inputs(seed=0):
    X = O.make_images(B, seed=seed)
    Z = O.make_latents(B, seed=5 + seed)
    eps = np.random.RandomState(6 + seed).randn(B, 100).astype(np.float32)
    return X, Z, eps


@pytest.fixture(scope="module")
def setup():
    import torch
    from neural_photo_editor_amd.trainer import Trainer
    P = make_train_params(O.make_params("IAN", 1))
    tr = Trainer(CFG, P, batch=B)
    tw = TrainTwin(P, dtype=torch.float64)
    return tr, tw, P


def dev(*arrs):
    import torch
    return [torch.from_numpy(np.ascontiguousarray(a)).cuda() for a in arrs]


def test_forward_losses_match_twin(setup):
    tr, tw, _ = setup
    X, Z, eps = inputs()
    tr.forward(*dev(X, Z, eps))
    m = tr.metrics()
    L = {k: float(v) for k, v in tw.losses(X, Z, eps).items()}
    for k in ("pixel_loss", "kl_div", "discrim_g_loss", "discrim_d_loss", "feature_loss", "gen_recon_loss", "gen_sample_loss",
              "pixel_acc", "discrim_acc"):
        assert abs(m[k] - L[k]) <= TOL_LOSS * max(1.0, abs(L[k])), (k, m[k], L[k])
    xh = tr.DZ["xhat"].cpu().numpy()
    assert rel(xh, tw.tensors["X_hat"].detach().numpy()) < 1e-4
    assert rel(tr.DG["xhat"].cpu().numpy(), tw.tensors["X_gen"].detach().numpy()) < 1e-4
    assert rel(tr.EX["p"].cpu().numpy(), tw.tensors["p_X"].detach().numpy()) < 1e-4


@pytest.mark.parametrize("which", ["gen", "discrim"])
def test_gradients_match_autograd(setup, which):
    tr, tw, _ = setup
    X, Z, eps = inputs(1)
    tr.forward(*dev(X, Z, eps))
    tr.backward(which)
    tr._regularizers(which)
    g, _ = tw.gradients(X, Z, eps)
    groups = ("dec", "Z") if which == "gen" else ("enc", "Z")
    worst = []
    for gname in groups:
        got = tr.grads_numpy(gname)
        for name, ref in g[gname].items():
            e = rel(got[name], ref.detach().numpy())
            worst.append((e, name))
    worst.sort(reverse=True)
    assert worst[0][0] < TOL_GRAD, worst[:8]


def test_two_updates_of_each_kind_track_the_twin(setup):
    """update_gen / update_discrim alternate as in train_IAN.py:497-504; parameters after 4 Adam steps."""
    import torch
    from neural_photo_editor_amd.trainer import Trainer
    _, _, P = setup
    tr = Trainer(CFG, P, batch=B)
    tw = TrainTwin(P, dtype=torch.float64)
    for it in range(4):
        X, Z, eps = inputs(10 + it)
        if it % 2 == 0:
            a = tr.update_gen(*dev(X, Z, eps)); b = tw.update_gen(X, Z, eps)
        else:
            a = tr.update_discrim(*dev(X, Z, eps)); b = tw.update_discrim(X, Z, eps)
        assert np.allclose(a, b, rtol=2e-3, atol=2e-4), (it, a, b)
    got, ref = tr.params_numpy(), tw.numpy_params()
    # Adam normalises the step: compare the parameter DISPLACEMENT against lr-sized steps
    worst = max((float(np.abs(got[n] - ref[n]).max()), n) for n in got)
    assert worst[0] < 0.25 * 4 * tr.lr, worst
