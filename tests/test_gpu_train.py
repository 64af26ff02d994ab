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
TOL_LOSS = 2e-5     # measured 1.4e-6 (tests/test_gpu_decomposition.py: loss_vs_plain_float64)
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


def twin32_noise(P, X, Z, eps):
    """How far a float32 run of the SAME restatement is from the float64 one: the conditioning of the comparison
    (leaky-ReLU / |.| kinks, batch statistics over 4 images).  -> {group: (median, max) of per-tensor rel. errors}"""
    import torch
    t64, t32 = TrainTwin(P, dtype=torch.float64), TrainTwin(P, dtype=torch.float32)
    g64, _ = t64.gradients(X, Z, eps)
    g32, _ = t32.gradients(X, Z, eps)
    out = {}
    for grp in g64:
        e = [rel(g32[grp][n].numpy(), g64[grp][n].numpy()) for n in g64[grp]]
        out[grp] = (float(np.median(e)), float(np.max(e)))
    return g64, out


def diverse_images(seed):
    """Batch whose samples differ strongly in contrast / brightness.  The MinibatchLayer differentiates
    |a_b - a_b'| (layers.py:507-511): for near-identical samples a float32 rounding flips sign(a_b - a_b') and
    changes the gradient at the 1e-3..1e-1 level; well separated samples keep the comparison well conditioned."""
    x = O.make_images(B, seed=seed)
    s = np.array([0.15, 0.45, 0.75, 1.0], np.float32)[:B].reshape(-1, 1, 1, 1)
    o = np.array([-0.6, 0.3, -0.1, 0.0], np.float32)[:B].reshape(-1, 1, 1, 1)
    return np.clip(x * s + o, -1, 1).astype(np.float32)


def test_encoder_passes_backward_sharp(setup):
    """Each encoder pass alone on well-separated images: CE seeds -> encoder_params gradients; CE + feature seeds
    -> d/d(image).  Well conditioned comparisons: must agree to float32 round-off."""
    import torch
    from oracle.train_twin import ENC_PARAMS
    from neural_photo_editor_amd.trainer import ENC_WIDTHS
    tr, tw, _ = setup
    X, Z, eps = inputs(2)
    imgs = [diverse_images(40), diverse_images(41)[::-1].copy(), np.roll(diverse_images(42), 1, 0)]
    tr.forward(*dev(imgs[0], Z, eps), xhat_override=dev(imgs[1])[0], xgen_override=dev(imgs[2])[0])
    enc = [tw.P[n] for n in ENC_PARAMS]
    N = tr.N
    t64 = lambda a: torch.tensor(a, dtype=torch.float64)
    feats = []
    for E, img, t in ((tr.EX, imgs[0], 0), (tr.EH, imgs[1], 1), (tr.EG, imgs[2], 2)):
        xin = t64(img).requires_grad_(True)
        g = tw.encoder(xin)
        p = tw.discriminator(g[3])
        feats.append((xin, g, p))
        ref = torch.autograd.grad((-torch.log(p[:, t])).mean(), enc, retain_graph=True)
        tr.enc_backward(E, (t, 1.0 / N, -1, 0.0), False, True, False, reset=True)
        got = tr.grads_numpy("enc")
        for n, r in zip(ENC_PARAMS, ref):
            assert rel(got[n], r.numpy()) < 2e-4, (t, n)
    # generator-side seeds into the image of the second pass: adv (target 0) + feature loss against the first pass
    xin, g, p = feats[1]
    gX = [a.detach() for a in feats[0][1]]
    floss = torch.stack([((a - b) ** 2).mean() for a, b in zip(gX, g)]).mean()
    (gx,) = torch.autograd.grad((-torch.log(p[:, 0])).mean() + floss, [xin])
    for i, w in enumerate(ENC_WIDTHS):
        cnt = (32 >> i) ** 2 * w
        tr.k.pair_loss(tr.EH["a%d" % (i + 1)], tr.EX["a%d" % (i + 1)], tr.EH["da%d" % (i + 1)], B * cnt, 1, 1, 1, 1.0 / (4.0 * N * cnt), 0,
                       tr.ws_loss, 1024, 0.0, tr.scalars[40:42])
    tr.enc_backward(tr.EH, (0, 1.0 / N, -1, 0.0), True, False, True, reset=True)
    got = tr.EH["dx"].cpu().numpy()[..., :3].transpose(0, 3, 1, 2)
    assert rel(got, gx.numpy()) < 2e-4


@pytest.mark.parametrize("which", ["gen", "discrim"])
def test_gradients_match_autograd(setup, which):
    """The composed update rules.  The encoder passes are fed the twin's X_hat / X_gen; what remains is the float32
    conditioning of the decoder chain, measured in the same test with a float32 run of the twin."""
    tr, tw, P = setup
    X, Z, eps = inputs(1)
    g, noise = twin32_noise(P, X, Z, eps)
    tw.losses(X, Z, eps)
    xh, xg = [t.detach().numpy().astype(np.float32) for t in (tw.tensors["X_hat"], tw.tensors["X_gen"])]
    tr.forward(*dev(X, Z, eps), xhat_override=dev(xh)[0], xgen_override=dev(xg)[0])
    tr.backward(which)
    tr._regularizers(which)
    for gname in (("dec", "Z") if which == "gen" else ("enc", "Z")):
        got = tr.grads_numpy(gname)
        errs = sorted(((rel(got[name], ref.numpy()), name) for name, ref in g[gname].items()), reverse=True)
        med, mx = float(np.median([e for e, _ in errs])), errs[0][0]
        _diag("gradients_match_autograd_%s_%s" % (which, gname), {"median": med, "max": mx, "twin32_median": noise[gname][0],
                                                                   "twin32_max": noise[gname][1], "worst": errs[:5]})
        # the float32 twin's error is ONE draw of a heavy-tailed quantity (batch statistics over 4 images, |.| kinks); the sharp
        # per-tensor form of this comparison is tests/test_gpu_decomposition.py.  Round 3: 8x / 8x with the float32 one-pass
        # variance; round 4 (float64 statistics): median within 2x, worst tensor within 4x of the twin's draw
        assert med < 2 * noise[gname][0] + 1e-5, (gname, med, noise[gname], errs[:5])
        assert mx < 4 * noise[gname][1] + 1e-3, (gname, mx, noise[gname], errs[:5])


def test_two_updates_of_each_kind_track_the_twin(setup):
    """update_gen / update_discrim alternate as in train_IAN.py:497-504.  (a) the reported losses follow the float64
    twin; (b) every Adam step is exactly lasagne.updates.adam (App. B.7) applied to the gradients the step itself
    computed, with one (t, m, v) state per group and Z_params stepped by BOTH updates (train_IAN.py:274-276)."""
    import torch
    from neural_photo_editor_amd.trainer import Trainer
    _, _, P = setup
    tr = Trainer(CFG, P, batch=B)
    tw = TrainTwin(P, dtype=torch.float64)
    ref = {g: {"p": grp.p.cpu().numpy().astype(np.float64), "m": np.zeros(grp.numel), "v": np.zeros(grp.numel), "t": 0}
           for g, grp in tr.groups.items()}
    for it in range(4):
        X, Z, eps = inputs(10 + it)
        gen = it % 2 == 0
        other = "enc" if gen else "dec"
        snap = tr.groups[other].p.clone()
        a = (tr.update_gen if gen else tr.update_discrim)(*dev(X, Z, eps))
        b = (tw.update_gen if gen else tw.update_discrim)(X, Z, eps)
        a, b = np.array(a), np.array(b)
        if not gen:
            a, b = np.delete(a, 2), np.delete(b, 2)      # discrim_acc is an argmax count over 12 decisions: not continuous
        # the first step sees identical parameters; later ones compare trajectories that Adam's sign-like first
        # steps let drift apart (float32 vs float64 sign of near-zero gradient entries)
        assert np.allclose(a, b, rtol=2e-3 if it == 0 else 3e-2, atol=2e-4), (it, a, b)
        for g in (("dec" if gen else "enc"), "Z"):
            r, grp = ref[g], tr.groups[g]
            grad = grp.g.cpu().numpy().astype(np.float64)
            r["t"] += 1
            a_t = tr.lr * np.sqrt(1 - 0.999 ** r["t"]) / (1 - 0.5 ** r["t"])
            r["m"] = 0.5 * r["m"] + 0.5 * grad
            r["v"] = 0.999 * r["v"] + 0.001 * grad * grad
            r["p"] = r["p"] - a_t * r["m"] / (np.sqrt(r["v"]) + 1e-8)
            assert grp.t == r["t"]
            assert np.abs(grp.p.cpu().numpy() - r["p"]).max() < 2e-6, (it, g)
        assert torch.equal(tr.groups[other].p, snap)                                # the other group is untouched
    got, twp = tr.params_numpy(), tw.numpy_params()
    moved = np.concatenate([(got[n] - np.asarray(P[n], np.float32)).ravel() for n in got])
    diff = np.concatenate([(got[n] - twp[n]).ravel() for n in got])
    assert np.abs(moved).mean() > 0.5 * tr.lr                       # parameters did move
    assert np.abs(diff).mean() < 0.5 * np.abs(moved).mean()         # and stay closer to the twin than to the start


def test_checkpoint_from_training_drives_the_inference_path(setup, tmp_path):
    """train -> GANcheckpoints-format npz (Theano names, running batch-norm averages, frozen MADE) -> API.IAN."""
    from neural_photo_editor_amd import IAN, checkpoints, config_loader, lowering
    from neural_photo_editor_amd.trainer import Trainer
    _, tw, P = setup
    tr = Trainer(CFG, P, batch=B)
    X, Z, eps = inputs(20)
    tr.update_gen(*dev(X, Z, eps))
    tr.update_discrim(*dev(X, Z, eps))
    state = tr.state_dict()
    # running averages: r <- 0.9 r + 0.1 batch, twice, from the real-data pass (the first step's batch statistics
    # are those of the twin's first forward)
    L = tw.losses(X, Z, eps)
    h2 = torch_conv_features(tw, X)
    m_batch = h2.mean((0, 2, 3)).detach().numpy()
    moved = state["bnorm2.mean"] - np.asarray(P["bnorm2.mean"])
    assert np.abs(moved).max() > 1e-3
    first = 0.9 * np.asarray(P["bnorm2.mean"], np.float64) + 0.1 * m_batch
    assert np.abs((state["bnorm2.mean"] - 0.9 * first) / 0.1 - m_batch).max() < 0.05      # second batch mean ~ first (tiny lr)
    fname = str(tmp_path / "IAN_trained.npz")
    tr.save_weights(fname, {"epoch": 3})
    specs = lowering.all_param_specs(config_loader.build_model(config_loader.load_config(CFG)))
    params, meta = checkpoints.load_weights(fname, specs)
    assert meta["epoch"] == 3 and abs(meta["learning_rate"] - tr.lr) < 1e-12
    model = IAN(CFG, True, params=params)
    z = O.make_latents(2, seed=9)
    ref = O.Oracle("IAN", state).sample_at(z)
    assert rel(model.sample_at(z), ref) < 1e-4
    x = O.make_images(2, seed=9)
    assert rel(model.encode_images(x), O.Oracle("IAN", state).encode_images(x)) < 1e-4


def _diag(name, obj):
    import json
    d = os.path.join(ROOT, "gpurun_out", "diag")
    try:
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, name + ".json"), "w") as fh:
            json.dump(obj, fh, indent=1)
    except OSError:
        pass


def torch_conv_features(tw, X):
    import torch
    import torch.nn.functional as F
    x = torch.tensor(X, dtype=torch.float64)
    h1 = F.leaky_relu(F.conv2d(x, tw.P["enc_conv1.W"], tw.P["enc_conv1.b"], stride=2, padding=2), 0.2)
    return F.conv2d(h1, tw.P["enc_conv2.W"], None, stride=2, padding=2)


def test_training_loop_end_to_end(setup, tmp_path):
    """train_IAN.py main-loop equivalent over the real step: 1 epoch, 2 chunks x 2 batches, checkpoint + metrics log."""
    from neural_photo_editor_amd import train_loop as TL, checkpoints
    from neural_photo_editor_amd.trainer import Trainer
    _, _, P = setup
    tr = Trainer(CFG, P, batch=B)
    cfg = dict(tr.cfg, batch_size=B, batches_per_chunk=2, max_epochs=1, shuffle=True)
    imgs = np.uint8((O.make_images(16, seed=77) + 1) * 127.5)
    wf = str(tmp_path / "IAN.npz")
    itr = TL.train(cfg, tr, TL.ArrayDataset(imgs), wf, to_device=lambda a: dev(a)[0])
    assert itr == 4 and tr.groups["dec"].t == 2 and tr.groups["enc"].t == 2 and tr.groups["Z"].t == 4
    recs = TL.read_records(str(tmp_path / "IANMETRICS.jsonl"))
    assert len(recs) == 2 and all(np.isfinite(v) for r in recs for v in r["metrics"].values())
    assert set(recs[0]["metrics"]) == set(TL.GEN_KEYS) | set(TL.DISCRIM_KEYS)
    with np.load(wf, allow_pickle=True) as f:
        assert "dec_conv1.W" in f.files and "bnorm2.inv_std" in f.files and "metadata" in f.files
    _, meta = checkpoints.load_weights(wf, [])
    assert meta["epoch"] == 0 and meta["itr"] == 4


def test_train_cli_runs_an_epoch(tmp_path):
    """python -m neural_photo_editor_amd.train_cli <config> --data ... : the train_IAN.py command line on the HIP path."""
    from neural_photo_editor_amd import train_cli, train_loop as TL
    src = open(CFG).read().replace("batch_size=16", "batch_size=4").replace("batches_per_chunk=64", "batches_per_chunk=2") \
        .replace("max_epochs=80", "max_epochs=1")
    assert "batch_size=4" in src and "batches_per_chunk=2" in src
    cfg_path = str(tmp_path / "IAN.py")
    open(cfg_path, "w").write(src)
    np.save(str(tmp_path / "imgs.npy"), np.uint8((O.make_images(16, seed=5) + 1) * 127.5))
    train_cli.main([cfg_path, "--data", str(tmp_path / "imgs.npy")])
    recs = TL.read_records(str(tmp_path / "IANMETRICS.jsonl"))
    assert len(recs) == 2 and recs[-1]["itr"] == 4 and os.path.exists(str(tmp_path / "IAN.npz"))
    train_cli.main([cfg_path, "--data", str(tmp_path / "imgs.npy"), "--resume", "--epochs", "2"])     # resumes at epoch 1
    assert len(TL.read_records(str(tmp_path / "IANMETRICS.jsonl"))) == 4
