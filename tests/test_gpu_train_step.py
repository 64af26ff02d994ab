"""ian_train_step (include/ian_train.h, csrc/ian_trainer.cpp): the whole train_IAN.py update behind one C entry, the ONLY
sequencer of the step since round 4 (trainer.Trainer is its thin ctypes caller).  The one-call entry is held bit for bit to
the same sequencer driven piece by piece -- the form tests/test_gpu_train*.py and tests/test_gpu_reference_pinned.py hold
against the float64 twin and the reference-executed fixtures -- and to the reference-executed metrics directly."""
import os

import numpy as np
import pytest

from neural_photo_editor_amd import synthetic as S

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CFG = os.path.join(ROOT, "neural_photo_editor_amd", "configs", "IAN.py")
GOLD = os.path.join(ROOT, "tests", "golden")
B = 4


def test_one_call_step_is_bitwise_the_piecewise_step():
    """ian_train_step (one call, HOST buffers, as a C caller would pass them) against the same sequencer driven piece by piece
    (ian_trainer_forward / metrics / backward / finish_allreduce / regularizers / apply_adam, DEVICE tensors): the parity tests
    of tests/test_gpu_train*.py and test_gpu_reference_pinned.py drive the pieces, the product calls the one entry -- they must
    be the same arithmetic bit for bit (metrics, every gradient, every parameter and running average after four updates)."""
    import torch
    from neural_photo_editor_amd.trainer import METRICS, Trainer
    P = S.make_train_params(S.make_params("IAN", 1))
    ct, tr = Trainer(CFG, P, B), Trainer(CFG, P, batch=B)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    for it, which in enumerate(("gen", "discrim", "gen", "discrim")):
        X, Z = S.make_images(B, seed=60 + it), S.make_latents(B, seed=70 + it)
        eps = np.random.RandomState(80 + it).randn(B, 100).astype(np.float32)
        mc = ct.step(which, X, Z, eps)
        tr.forward(dev(X), dev(Z), dev(eps))
        mp = tr.metrics()
        tr.backward(which)
        tr._finish_allreduce(which)
        tr._regularizers(which)
        for k in METRICS:
            assert mc[k] == mp[k], (it, k, mc[k], mp[k])
        for gname in (("dec" if which == "gen" else "enc"), "Z"):
            for name, g in tr.grads_numpy(gname).items():
                assert np.array_equal(ct.read(name, grad=True), g), (it, name)
        tr._apply_adam(which)
    for name, v in tr.state_dict().items():
        if not name.startswith("l_IAF_"):
            assert np.array_equal(ct.read(name), v), name
    assert ct.adam_steps() == (2, 4, 2) == tr.adam_steps()     # encoder_params, Z_params (both updates), decoder_params
    moved = np.abs(ct.read("dec_conv1.W") - P["dec_conv1.W"]).mean()
    assert moved > 0.2 * tr.lr
    assert np.abs(ct.read("bnorm2.mean") - P["bnorm2.mean"]).max() > 1e-3     # running averages follow the real-data pass


def test_step_rejects_mistyped_buffers():
    """a non-contiguous / float64 / mis-shaped tensor must fail loudly, not train on garbage."""
    import torch
    from neural_photo_editor_amd.trainer import Trainer, IanTrainError
    P = S.make_train_params(S.make_params("IAN", 1))
    tr = Trainer(CFG, P, batch=B)
    X, Z = torch.zeros(B, 3, 64, 64, device="cuda"), torch.zeros(B, 100, device="cuda")
    for bad in ((X.double(), Z, Z), (X.permute(0, 1, 3, 2), Z, Z), (X, Z[:, :50], Z), (X, Z, torch.zeros(B, 128, device="cuda")[:, :100]),
                (X[:2], Z[:2], Z[:2])):
        with pytest.raises(IanTrainError):
            tr.step("gen", *bad)


def test_c_step_metrics_vs_reference_train_IAN_and_device_pointers():
    import torch
    from neural_photo_editor_amd.trainer import GEN_KEYS, Trainer
    fx = np.load(os.path.join(GOLD, "ref_train_IAN.npz"))
    P = S.make_train_params(S.make_params("IAN", 1))
    ct = Trainer(CFG, P, int(fx["batch"]))
    b = int(fx["batch"])
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    got = ct.update_gen(dev(fx["X"][:b]), dev(fx["Z"][:b]), dev(fx["gen/eps"]))                # device pointers
    assert fx["gen/metric_names"].tolist() == list(GEN_KEYS)
    assert np.allclose(got, fx["gen/metrics"], rtol=2e-4, atol=2e-4), (got, fx["gen/metrics"])
    with pytest.raises(Exception):
        ct.step("gen", fx["X"][:2], fx["Z"][:2], fx["gen/eps"][:2])                           # wrong batch: loud


def test_c_step_from_host_buffers_only():
    """No torch on the device anywhere: numpy in, nine floats out, parameters read back -- what a C caller does.  (Also the
    scenario tests/test_sanitize.py runs against the ASan/UBSan build: torch's own HIP initialisation does not survive an ASan
    preload on this image, libian's does.)"""
    from neural_photo_editor_amd.trainer import DISCRIM_KEYS, GEN_KEYS, Trainer
    fx = np.load(os.path.join(GOLD, "ref_train_IAN.npz"))
    b = int(fx["batch"])
    P = S.make_train_params(S.make_params("IAN", 1))
    ct = Trainer(CFG, P, b)
    got = ct.update_gen(fx["X"][:b], fx["Z"][:b], fx["gen/eps"])
    assert np.allclose(got, fx["gen/metrics"], rtol=2e-4, atol=2e-4), (got, fx["gen/metrics"])
    got = np.array(ct.update_discrim(fx["X"][b:], fx["Z"][b:], fx["discrim/eps"]))
    keep = [i for i, n in enumerate(DISCRIM_KEYS) if n != "discrim_acc"]
    assert np.allclose(got[keep], fx["discrim/metrics"][keep], rtol=1e-2, atol=1e-3), (got, fx["discrim/metrics"])
    assert ct.adam_steps() == (1, 2, 1)
    assert np.abs(ct.read("enc_fc1.W") - P["enc_fc1.W"]).max() > 0            # Z_params moved (both updates)
    assert np.array_equal(ct.read("bnorm2.mean", grad=False).shape, P["bnorm2.mean"].shape)
    ct.close()                                                                  # guard bands are verified here in the sanitized build


@pytest.mark.parametrize("which,B", [("gen", 4), ("discrim", 4), ("gen", 32)])
def test_weight_gradient_stream_is_bitwise_the_single_stream_step(which, B):
    """overlap_wgrad=1 (default) issues every weight-gradient GEMM on a second stream, joined before the regularizers.  The
    COLD first step is the sharp case: each layer builds its split-K schedule and zeroes its partial buffer on first use, and that
    zeroing must have landed before the second stream's GEMM writes the same buffer."""
    from neural_photo_editor_amd.trainer import Trainer
    P = S.make_train_params(S.make_params("IAN", 1))
    X, Z = S.make_images(B, seed=60), S.make_latents(B, seed=70)
    eps = np.random.RandomState(80).randn(B, 100).astype(np.float32)
    runs = {}
    for ov in (0, 1):
        ct = Trainer(CFG, P, B)
        ct.set_option("overlap_wgrad", ov)
        for _ in range(2):                                     # cold step, then a warm one on the updated parameters
            ct.step(which, X, Z, eps)
        names = [n for n in ct.shapes if not n.startswith("l_IAF")]
        runs[ov] = ({n: ct.read(n) for n in names},
                    {n: ct.read(n, grad=True) for n in names if not n.endswith((".mean", ".inv_std"))})
        ct.close()
    for kind in (0, 1):
        for n, v in runs[0][kind].items():
            assert np.array_equal(runs[1][kind][n], v), (("param", "grad")[kind], n)
