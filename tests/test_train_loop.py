"""CPU tests of the training-loop plumbing (train_IAN.py:357-571 equivalents) and the NPE host-side steps."""
import json
import os

import numpy as np

from neural_photo_editor_amd import npe_ops, train_loop as TL

CFG = dict(batch_size=4, batches_per_chunk=3, num_latents=100, update_ratio=1, shuffle=True, max_epochs=3, checkpoint_every_nth=1,
           learning_rate={0: 0.0002, 1: 0.0001, 2: 0.00005}, decay_rate=0, seed=0)


class FakeTrainer:
    def __init__(self):
        self.lr, self.calls, self.saved = 0.0002, [], []

    def update_gen(self, X, Z, eps):
        assert X.shape == (4, 3, 64, 64) and Z.shape == (4, 100) and eps.shape == (4, 100) and X.dtype == np.float32
        assert -1.0 <= X.min() and X.max() <= 1.0
        self.calls.append("g")
        return [1.0, 2.0, 3.0, 4.0, 5.0]

    def update_discrim(self, X, Z, eps):
        self.calls.append("d")
        return [6.0, 7.0, 8.0, 9.0, 10.0]

    def save_weights(self, fname, metadata):
        self.saved.append((fname, dict(metadata)))


def dataset(n=30):
    rs = np.random.RandomState(0)
    return TL.ArrayDataset(rs.randint(0, 256, (n, 3, 64, 64)).astype(np.uint8))


def test_data_loader_chunks_and_seeding():
    ds = dataset(30)
    chunks = list(TL.data_loader(CFG, ds, offset=0, shuffle=True, seed=3))
    assert len(chunks) == 30 // 12 and chunks[0].shape == (12, 3, 64, 64) and chunks[0].dtype == np.float32
    again = list(TL.data_loader(CFG, ds, offset=0, shuffle=True, seed=3))
    assert all(np.array_equal(a, b) for a, b in zip(chunks, again))
    other = list(TL.data_loader(CFG, ds, offset=0, shuffle=True, seed=4))
    assert not np.array_equal(chunks[0], other[0])
    perm = np.random.RandomState(3).permutation(30)
    assert np.array_equal(chunks[0], TL.to_tanh(ds.images[perm[:12]]))           # train_IAN.py:363,371
    plain = list(TL.data_loader(CFG, ds, offset=2, shuffle=False))
    assert np.array_equal(plain[0], TL.to_tanh(ds.images[2:14]))


def test_learning_rate_schedule():
    assert TL.learning_rate_for(CFG, 0, 0.0002) == 0.0002          # epoch 0 never changes it (train_IAN.py:442)
    assert TL.learning_rate_for(CFG, 1, 0.0002) == 0.0001
    assert TL.learning_rate_for(CFG, 5, 0.0001) == 0.0001
    c = dict(CFG, learning_rate=0.001, decay_rate=0.1)
    assert abs(TL.learning_rate_for(c, 2, 0.001) - 0.0009) < 1e-12


def test_training_loop_alternation_metrics_checkpoints_resume(tmp_path):
    tr = FakeTrainer()
    wf = str(tmp_path / "IAN.npz")
    itr = TL.train(CFG, tr, dataset(30), wf)
    assert itr == 3 * 2 * 3                                          # epochs x chunks x batches
    assert tr.calls == ["g", "d"] * 9                                 # strict alternation across chunks (train_IAN.py:497-504)
    assert [m["epoch"] for _, m in tr.saved] == [0, 1, 2] and tr.saved[-1][1]["itr"] == 18
    assert abs(float(tr.saved[1][1]["learning_rate"]) - 0.0001) < 1e-9 and abs(tr.lr - 0.00005) < 1e-12
    recs = TL.read_records(str(tmp_path / "IANMETRICS.jsonl"))
    assert len(recs) == 6 and recs[0]["metrics"]["gen_recon_loss"] == 1.0 and recs[0]["metrics"]["discrim_acc"] == 8.0
    assert "_stamp" in recs[0] and recs[-1]["itr"] == 18
    # resume: continues after the stored epoch with the stored learning rate, appends to the log
    tr2 = FakeTrainer()
    TL.train(CFG, tr2, dataset(30), wf, resume=True, load_metadata={"epoch": 1, "learning_rate": 0.0001})
    assert [m["epoch"] for _, m in tr2.saved] == [2] and abs(tr2.lr - 0.00005) < 1e-12
    assert len(TL.read_records(str(tmp_path / "IANMETRICS.jsonl"))) == 8


def test_sample_grid_layout():
    rs = np.random.RandomState(1)
    sample = lambda z: np.tanh(z[:, :1, None, None] * np.ones((1, 3, 64, 64), np.float32))
    zfn = lambda x: x.reshape(len(x), -1)[:, :100].astype(np.float32)
    endpoints = rs.randint(0, 256, (6, 3, 64, 64)).astype(np.uint8)
    imgs = TL.sample_grid(sample, zfn, endpoints, 100, rs)
    assert imgs.shape == (54, 3, 64, 64) and imgs.dtype == np.uint8
    assert np.array_equal(imgs[27], endpoints[0]) and np.array_equal(imgs[35], endpoints[1])   # [endpoint, 7 interpolants, endpoint]
    Z = TL.interpolation_latents(zfn(TL.to_tanh(endpoints)))
    assert Z.shape == (21, 100) and np.allclose(Z[0], zfn(TL.to_tanh(endpoints))[0]) and np.allclose(Z[6], zfn(TL.to_tanh(endpoints))[1])
    grid = TL.tile_grid(imgs)
    assert grid.shape == (6 * 64, 9 * 64, 3) and np.array_equal(grid[:64, 64:128].transpose(2, 0, 1), imgs[1])


def test_npe_host_steps():
    class M:
        def imgradRGB(self, x1, y1, x2, y2, rgb, z):
            assert rgb.shape == (1, 3, 64, 64) and z.shape == (1, 100)
            return np.ones((1, 100), np.float32)

        def imgrad(self, x1, y1, x2, y2, z):
            return np.full((1, 100), 2.0, np.float32)

        def sample_at(self, z):
            return np.full((1, 3, 64, 64), 0.5, np.float32)
    Z = np.zeros((10, 10), np.float32)
    Z2 = npe_ops.brush_step(M(), Z, (26, 26, 30, 30), np.zeros((3, 64, 64), np.uint8))
    assert Z2.shape == (10, 10) and np.allclose(Z2, -0.05 * 5)                   # NPE.py:199-209: weight*grad*(1+(x2-x1))
    assert np.allclose(npe_ops.lighten_step(M(), Z, (0, 0, 4, 4)), 0.1 * 2.0 * 5)  # NPE.py:311-312: weight*grad*(1+(x2-x1))
    assert np.allclose(npe_ops.lighten_step(M(), Z, (0, 0, 4, 4), sign=-1.0), -1.0)
    recon = np.full((3, 64, 64), 127, np.uint8)
    im, mask = npe_ops.photo_blend(M(), Z, recon, np.zeros((3, 64, 64), np.float32))
    delta = np.float32(0.5) - npe_ops.to_tanh(np.float32(recon))
    assert mask.dtype == np.float64 and np.allclose(mask, np.abs(delta).mean(0), atol=1e-6)   # constant field: unchanged by the Gaussian
    assert im.dtype == np.uint8 and np.array_equal(im, np.uint8(npe_ops.from_tanh(npe_ops.to_tanh(recon) + mask * delta)))


def test_paint_event_composes_the_reference_lines():
    """npe_ops.paint_event without a device model = NPE.py:204-231 line for line; the latent update follows the reference's
    order of float32 operations (grad = temp*(1+(x2-x1)); Z -= weight*grad), which is what ian_brush_step reproduces."""
    rs = np.random.RandomState(5)
    G = rs.randn(1, 100).astype(np.float32)

    class M:
        def imgradRGB(self, x1, y1, x2, y2, rgb, z):
            return G

        def sample_at(self, z):
            return np.tanh(np.float32(z[:, :1, None, None]) + np.zeros((1, 3, 64, 64), np.float32))
    Z = rs.randn(10, 10).astype(np.float32)
    box = (20, 21, 27, 28)
    Zn, x = npe_ops.paint_event(M(), Z, box, np.zeros((3, 64, 64), np.uint8))
    grad = G[0].reshape(10, 10) * (1 + (box[2] - box[0]))                       # NPE.py:206 (float32 * python int)
    want = Z - np.float32(0.05) * grad                                         # NPE.py:209
    assert Zn.dtype == np.float32 and np.array_equal(Zn, want)
    assert np.array_equal(x, M().sample_at(want.reshape(1, -1))[0])
    recon = rs.randint(0, 256, (3, 64, 64)).astype(np.uint8)
    err = (rs.randn(3, 64, 64) * 0.05).astype(np.float32)
    Zp, im = npe_ops.paint_event(M(), Z, box, np.zeros((3, 64, 64), np.uint8), recon, err)
    assert np.array_equal(Zp, want)
    assert np.array_equal(im, npe_ops.photo_blend_host(M().sample_at(want.reshape(1, -1))[0], recon, err)[0])


def test_photo_blend_host_is_the_reference_expression():
    """NPE.py:218-231 written out independently (scipy's gaussian_filter, float64 mask, bare uint8 cast) == photo_blend_host,
    and the separable restatement the HIP kernel follows == scipy bit for bit (also on out-of-range data, which wraps)."""
    import scipy.ndimage
    rs = np.random.RandomState(3)
    for scale in (0.3, 1.0, 2.5):
        RECON = rs.randint(0, 256, (3, 64, 64)).astype(np.uint8)
        IMG = rs.randint(0, 256, (3, 64, 64)).astype(np.uint8)
        ERROR = npe_ops.to_tanh(np.float32(IMG)) - npe_ops.to_tanh(np.float32(RECON))            # NPE.py:263
        xhat = np.clip(npe_ops.to_tanh(np.float32(RECON)) + scale * rs.randn(3, 64, 64).astype(np.float32) * 0.3, -1, 1).astype(np.float32)
        DELTA = xhat - npe_ops.to_tanh(np.float32(RECON))
        MASK = scipy.ndimage.gaussian_filter(np.min([np.mean(np.abs(DELTA), axis=0), np.ones((64, 64))], axis=0), 0.7)
        D = MASK * DELTA + (1 - MASK) * ERROR
        with np.errstate(invalid="ignore"):
            IM = np.uint8(npe_ops.from_tanh(npe_ops.to_tanh(RECON) + D))
        im, mask = npe_ops.photo_blend_host(xhat, RECON, ERROR)
        assert np.array_equal(im, IM) and np.array_equal(mask, MASK)
        m0 = np.min([np.mean(np.abs(DELTA), axis=0), np.ones((64, 64))], axis=0)
        assert np.array_equal(npe_ops.separable_reflect_filter(m0, npe_ops.gaussian_half_kernel()), MASK)
    assert npe_ops.gaussian_half_kernel().shape == (4,)


def test_train_cli_arguments_and_data_loading(tmp_path):
    from neural_photo_editor_amd import train_cli
    a = train_cli.parse_args(["cfg/IAN.py", "--resume", "--epochs", "2", "--batch", "8"])
    assert a.config_path == "cfg/IAN.py" and a.resume and a.epochs == 2 and a.batch == 8 and not a.local_statistics
    assert train_cli.load_images(None, 6).shape == (6, 3, 64, 64)
    imgs = np.random.RandomState(0).randint(0, 256, (5, 3, 64, 64)).astype(np.uint8)
    np.save(str(tmp_path / "d.npy"), imgs)
    np.savez(str(tmp_path / "d.npz"), images=imgs)
    assert np.array_equal(train_cli.load_images(str(tmp_path / "d.npy")), imgs)
    assert np.array_equal(train_cli.load_images(str(tmp_path / "d.npz")), imgs)
    np.save(str(tmp_path / "bad.npy"), imgs.astype(np.float32))
    try:
        train_cli.load_images(str(tmp_path / "bad.npy"))
        assert False
    except ValueError:
        pass


def test_fresh_run_starts_from_the_config_initialisers():
    """train_IAN.py starts from lasagne's initial values (IAN.py: Normal(0.02) filters, gamma 1, beta 0, running mean 0 /
    inv_std 1, Orthogonal('relu') MADE, theta ~ N(0, 0.05), minibatch b = -1), not from the perturbed test fixtures."""
    import os
    from neural_photo_editor_amd import train_cli, config_loader as cl, lowering
    from neural_photo_editor_amd import synthetic
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    mod = cl.load_config(os.path.join(root, "neural_photo_editor_amd", "configs", "IAN.py"))
    specs = lowering.all_param_specs(cl.build_model(mod))
    P = train_cli.initial_params(specs, 0)
    want = dict(synthetic.param_shapes("IAN"), **synthetic.train_param_shapes())
    assert {k: v.shape for k, v in P.items()} == {k: tuple(v) for k, v in want.items()}
    assert np.all(P["bnorm2.gamma"] == 1) and np.all(P["bnorm2.beta"] == 0) and np.all(P["bnorm2.mean"] == 0) and np.all(P["bnorm2.inv_std"] == 1)
    assert np.all(P["minibatch_discrim.b"] == -1) and np.all(P["minibatch_discrim.log_weight_scale"] == 0)
    assert abs(P["enc_conv2.W"].std() - 0.02) < 1e-3 and abs(P["minibatch_discrim.theta"].std() - 0.05) < 2e-3
    W = P["l_IAF_mu_input.W"].astype(np.float64)
    assert np.allclose(W @ W.T, 2.0 * np.eye(100), atol=1e-4)                 # Orthogonal('relu'): gain sqrt(2)
    assert np.allclose(P["R_coeff_base"], 0.25) and np.all(P["enc_conv1.b"] == 0)
    Q = train_cli.initial_params(specs, 0)
    assert all(np.array_equal(P[k], Q[k]) for k in P)                         # every rank draws the same values
