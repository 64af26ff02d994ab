"""REFERENCE-EXECUTED section: the CPU oracle (and the product's host-side mask generator) against fixtures that
the reference's OWN files computed (tests/golden/ref_*.npz, written by tests/golden/make_ref_golden.py by running
/root/reference/{layers,mask_generator,IAN,IAN_simple,API,GANcheckpoints,train_IAN,sample_IAN}.py unmodified on the
evaluating Theano/Lasagne stand-in of oracle/refexec/).  This is what pins the oracle: the restatement in oracle/ and
the reference-owned arithmetic agree to float64 round-off.  Still [recalled]: third-party primitive conventions
(conv / transposed conv / dilated conv / batch_norm / adam), see oracle/refexec/minilasagne.py.

When /root/reference is present (build container) one test re-executes the reference and checks the committed
fixtures are what it produces today; on the GPU box that test is skipped, the fixtures travel."""
import os

import numpy as np
import pytest
import torch

from oracle import ian_oracle as O
from oracle.torch_twin import TorchTwin
from oracle import train_twin as TT

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
F64 = 1e-9      # float64 restatement vs float64 reference execution
F32 = 1e-4      # float32 restatement vs float64 reference execution (north-star tolerance)


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


def load(name):
    return np.load(os.path.join(GOLD, name))


def sub(fx, prefix):
    """{'m/W': ..} -> {'mW': ..}: parameters of one small-layer fixture under their Theano names."""
    n = len(prefix) + 1
    return {k[n:]: fx[k].astype(np.float64) for k in fx.files if k.startswith(prefix + "/")}


# ---- MADE masks: produced by mask_generator.MaskGenerator itself ---------------------------------------------
def test_made_masks_from_the_reference_mask_generator():
    g = load("ref_made_masks.npz")
    M = np.unpackbits(g["packed"])[:3 * 100 * 100].reshape(3, 100, 100)
    assert g["counts"].tolist() == [100, 9900, 4950] and int(g["child_seed"]) == 822569775
    ordering, child = O.made_ordering()
    assert child == int(g["child_seed"]) and np.array_equal(ordering, g["ordering"])
    for mine, theirs in zip(O.made_masks(), M):
        assert np.array_equal(mine.astype(np.uint8), theirs)
    # the product's host-side generator (neural_photo_editor_amd/made.py) -- bit-exact, SURVEY a18
    from neural_photo_editor_amd import made
    for mine, theirs in zip(made.masks_once(100), M):
        assert np.array_equal(np.asarray(mine).astype(np.uint8), theirs)


# ---- layers.py building blocks ---------------------------------------------------------------------------------
@pytest.mark.parametrize("tag,scales", [("mdcl_02", [0, 2]), ("mdcl_023", [0, 2, 3]), ("mdcl_234", [2, 3, 4])])
def test_mdcl_vs_reference(tag, scales):
    fx = load("ref_layers.npz")
    P = sub(fx, tag)
    y = O.mdcl(P["x"], P, "m", scales)
    assert rel(y, fx[tag + "/y"]) < F64


def test_mdblock_vs_reference_both_modes():
    fx = load("ref_layers.npz")
    P = sub(fx, "mdblock")
    orc = O.Oracle.__new__(O.Oracle)
    orc.P = P
    assert rel(orc.mdblock(P["x"], "blk", [0, 2, 3]), fx["mdblock/y_det"]) < F64
    tw = TT.TrainTwin.__new__(TT.TrainTwin)           # batch-statistics BN (train_IAN.py passes)
    tw.P = {k: torch.tensor(v) for k, v in P.items()}
    y = tw.mdblock(tw.P["x"], "blk", [0, 2, 3]).numpy()
    assert rel(y, fx["mdblock/y_train"]) < F64


def test_beta_deconv_minibatch_gauss_iaf_vs_reference():
    fx = load("ref_layers.npz")
    assert rel(O.beta_layer(fx["beta/a"].astype(np.float64), fx["beta/b"].astype(np.float64)), fx["beta/y"]) < F64
    x, W = fx["deconv/x"].astype(np.float64), fx["deconv/dc.W"].astype(np.float64)
    assert rel(O.deconv5s2(x, W, None, flip=True), fx["deconv/y"]) < F64
    assert rel(O.deconv5s2(x, W, None, flip=False), fx["deconv/y"]) > 0.1      # the flip is not a no-op
    tw = TT.TrainTwin.__new__(TT.TrainTwin)
    tw.P = {"minibatch_discrim." + k[3:]: torch.tensor(v) for k, v in sub(fx, "minibatch").items() if k.startswith("mb.")}
    y = tw.minibatch(torch.tensor(fx["minibatch/x"].astype(np.float64))).numpy()
    assert rel(y, fx["minibatch/y"]) < F64
    mu, ls, eps = (fx["gauss/" + k].astype(np.float64) for k in ("mu", "ls", "eps"))
    assert np.array_equal(fx["gauss/y_det"], mu)
    assert rel(mu + np.exp(ls) * eps, fx["gauss/y"]) < F64
    g = load("ref_made_masks.npz")
    masks = tuple(m.astype(np.float64) for m in np.unpackbits(g["packed"])[:30000].reshape(3, 100, 100))
    P = sub(fx, "iaf")
    z = P["z"]
    # get_output(MADE) feeds the masked MLP with its own first hidden layer (layers.py:775 overwrites input_layer)
    m, s = O.made_as_wired(z, P, "l_IAF_mu", masks), O.made_as_wired(z, P, "l_IAF_ls", masks)
    assert rel(m, fx["iaf/made_mu"]) < F64 and rel(s, fx["iaf/made_ls"]) < F64
    assert rel((z - m) / np.exp(s), fx["iaf/y"]) < F64
    # ... while the masked MLP itself (MADE.final_layer fed with z) is the textbook MADE
    assert rel(O.made(z, P, "l_IAF_mu", masks), fx["iaf/final_layer_of_z"]) < F64
    assert rel(O.made(z, P, "l_IAF_mu", masks), fx["iaf/made_mu"]) > 0.1


# ---- API.IAN: encode / decode / brush gradients ---------------------------------------------------------------
@pytest.mark.parametrize("arch", O.ARCHS)
def test_oracle_inference_vs_reference_api(arch):
    fx = load("ref_%s.npz" % arch)
    P = O.make_params(arch, 1)
    x, zs = fx["x"], fx["z_sample"]
    assert np.array_equal(x, O.make_images(2, 0)) and np.array_equal(zs, O.make_latents(2, 2))
    for dtype, tol in ((np.float64, 1e-6), (np.float32, F32)):   # 1e-6: the fixture stores images in float32
        orc = O.Oracle(arch, P, dtype=dtype)
        assert rel(orc.Zfn(x), fx["zpre"]) < tol
        assert rel(orc.encode_images(x), fx["z"]) < tol
        assert rel(orc.sample_at(fx["z"]), fx["xhat"]) < tol
        assert rel(orc.sample_at(zs), fx["x_sample"]) < tol
        if arch == "IAN":
            assert rel(orc.Z_IAF_fn(zs), fx["z_iaf_of_sample"]) < tol
            assert rel(orc.sample(zs), fx["x_from_ziaf"]) < tol
    orc = O.Oracle(arch, P, dtype=np.float64)
    for i, f in enumerate(orc.encoder_features(x[:1])):
        k = "enc_conv%d" % (i + 1)
        stride = int(fx["actstat_" + k][0])
        assert rel(f.ravel()[::stride], fx["act_" + k]) < F64 and rel(f.sum(), fx["actstat_" + k][1]) < 1e-8
    checked = 0
    for name, a in orc.decoder_activations(zs[:1]):
        if "act_" + name in fx.files:
            stride = int(fx["actstat_" + name][0])
            assert rel(a.ravel()[::stride], fx["act_" + name]) < F64, name
            assert rel(np.abs(a).sum(), fx["actstat_" + name][2]) < 1e-8, name
            checked += 1
    assert checked == (4 if arch == "IAN_simple" else 8)


@pytest.mark.parametrize("arch", O.ARCHS)
def test_oracle_brush_gradients_vs_reference_api(arch):
    fx = load("ref_%s.npz" % arch)
    tw = TorchTwin(arch, O.make_params(arch, 1), dtype=torch.float64)
    rgb = np.full((1, 3, 64, 64), -1.0, np.float32)
    rgb[:, 0] = 1.0
    z = fx["z_sample"][:1]
    for k, (c1, r1, c2, r2) in enumerate(fx["patches"].tolist()):
        assert rel(tw.imgradRGB(c1, r1, c2, r2, rgb, z), fx["grad_rgb_%d" % k]) < 1e-8
        assert rel(tw.imgrad(c1, r1, c2, r2, z), fx["grad_light_%d" % k]) < 1e-8
    # the committed oracle-generated goldens agree with the reference-executed ones
    g = load("%s_seed1.npz" % arch)
    assert rel(g["grad_rgb"], fx["grad_rgb_0"]) < 1e-6 and rel(g["grad_light"], fx["grad_light_0"]) < 1e-6
    assert rel(g["z"], fx["z"]) < F32 and rel(g["xhat"], fx["xhat"]) < F32
    # 10 steps of NPE.py:199-209 with the twin's gradient
    Z = z.astype(np.float64).copy()
    c1, r1, c2, r2 = fx["patches"][0].tolist()
    for _ in range(10):
        Z = (Z - 0.05 * tw.imgradRGB(c1, r1, c2, r2, rgb, Z.astype(np.float32)) * (1 + (c2 - c1))).astype(np.float32).astype(np.float64)
    assert rel(Z, fx["z_after_10_brush_steps"]) < 1e-6


# ---- train_IAN.make_training_functions -----------------------------------------------------------------------
def _grad_ok(fx, tag, name, g, tol):
    g = np.asarray(g, np.float64)
    if "%s/grad/%s" % (tag, name) in fx.files:
        ref = fx["%s/grad/%s" % (tag, name)]
        return rel(g, ref) < tol, rel(g, ref)
    st = fx["%s/grad_stat/%s" % (tag, name)]
    e1 = rel(g.ravel()[::int(st[0])], fx["%s/grad_sample/%s" % (tag, name)])
    e2 = abs(np.sqrt((g * g).sum()) - st[3]) / (st[3] + 1e-30)
    return (e1 < tol and e2 < tol), max(e1, e2)


def test_train_twin_vs_reference_training_functions():
    fx = load("ref_train_IAN.npz")
    B = int(fx["batch"])
    P = TT.make_train_params(O.make_params("IAN", 1))
    tw = TT.TrainTwin(P, dtype=torch.float64)
    X, Z = fx["X"], fx["Z"]
    assert np.array_equal(X, O.make_images(2 * B, 21))
    # parameter groups are the reference's (train_IAN.py:184-194)
    assert sorted(fx["gen/params"].tolist()) == sorted(tw.groups["dec"] + tw.groups["Z"])
    assert sorted(fx["discrim/params"].tolist()) == sorted(tw.groups["enc"] + tw.groups["Z"])
    assert all(n.startswith("l_IAF") or n.endswith(".mean") or n.endswith(".inv_std") for n in fx["untrained"].tolist())
    # itr 0: update_gen on batch 0 -- metrics and every gradient
    g, _ = tw.gradients(X[:B], Z[:B], fx["gen/eps"])
    worst = 0.0
    for grp in ("dec", "Z"):
        for name, gv in g[grp].items():
            ok, e = _grad_ok(fx, "gen", name, gv.numpy(), 1e-7)
            assert ok, (name, e)
            worst = max(worst, e)
    m = tw.update_gen(X[:B], Z[:B], fx["gen/eps"])
    assert fx["gen/metric_names"].tolist() == ["gen_recon_loss", "gen_sample_loss", "pixel_loss", "feature_loss", "pixel_acc"]
    assert rel(m, fx["gen/metrics"]) < 1e-9
    # itr 1: update_discrim on batch 1, after the gen update
    g, _ = tw.gradients(X[B:], Z[B:], fx["discrim/eps"])
    for grp in ("enc", "Z"):
        for name, gv in g[grp].items():
            ok, e = _grad_ok(fx, "discrim", name, gv.numpy(), 1e-6)
            assert ok, (name, e)
    m = tw.update_discrim(X[B:], Z[B:], fx["discrim/eps"])
    assert fx["discrim/metric_names"].tolist() == ["discrim_g_loss", "discrim_d_loss", "discrim_acc", "pixel_loss", "pixel_acc"]
    assert rel(m, fx["discrim/metrics"]) < 1e-8
    # update_discrim taken from the INITIAL parameters (the fixture's 'discrim0'): the step the GPU test holds tight
    tw0 = TT.TrainTwin(P, dtype=torch.float64)
    g, _ = tw0.gradients(X[:B], Z[:B], fx["discrim0/eps"])
    for grp in ("enc", "Z"):
        for name, gv in g[grp].items():
            ok, e = _grad_ok(fx, "discrim0", name, gv.numpy(), 1e-7)
            assert ok, (name, e)
    assert rel(tw0.update_discrim(X[:B], Z[:B], fx["discrim0/eps"]), fx["discrim0/metrics"]) < 1e-9
    # parameters after the two Adam updates (Z group stepped twice by ONE Adam instance, train_IAN.py:266-276)
    # (Adam's first steps are m/(sqrt(v)+1e-8) ~ sign(g): entries with |g| ~ 1e-8 amplify round-off, hence 2e-6)
    for name in tw.groups["dec"] + tw.groups["Z"] + tw.groups["enc"]:
        v = tw.P[name].detach().numpy()
        if "after/" + name in fx.files:
            assert rel(v, fx["after/" + name]) < 2e-6, name
        else:
            ref = fx["after_sample/" + name]
            stride = max(1, -(-v.size // 1024))
            stride += (stride > 1 and stride % 2 == 0)
            assert rel(v.ravel()[::stride], ref) < 2e-6, name


# ---- one NPE.py editing session (facade-call sequence, no UI) -------------------------------------------------------------------
class _TwinFacade:
    """API.py's method names over the float64 torch twin of the oracle (a test double for tests/session_replay.py)."""

    def __init__(self, tw):
        self.tw = tw

    def encode_images(self, x):
        return self.tw.np_encode(np.asarray(x, np.float32)).astype(np.float32)

    def sample_at(self, z):
        return self.tw.np_decode(np.asarray(z, np.float32)).astype(np.float32)

    def imgradRGB(self, c1, r1, c2, r2, rgb, z):
        return np.asarray(self.tw.imgradRGB(c1, r1, c2, r2, np.asarray(rgb, np.float32), np.asarray(z, np.float32)), np.float64).astype(np.float32)

    def imgrad(self, c1, r1, c2, r2, z):
        return np.asarray(self.tw.imgrad(c1, r1, c2, r2, np.asarray(z, np.float32)), np.float64).astype(np.float32)


def test_npe_session_replay_oracle_vs_reference_executed_session():
    """tests/golden/ref_session_IAN_simple.npz: infer -> 6 brush events in photo mode (two colours, two brush sizes) -> 3 scroll events ->
    Reset, as NPE.py's callbacks sequence the reference's own API.IAN (NPE.py:192-235, 239-279, 305-316, 330-340).  The oracle's float64
    twin, wrapped in API.py's method names and driven through npe_ops' host composition, must land on the same latents, blend masks and
    canvas bytes: this pins the oracle (and npe_ops' restatement of the lines between the model calls) to the executed reference."""
    import torch
    from session_replay import replay, compare
    fx = load("ref_session_IAN_simple.npz")
    tw = TorchTwin("IAN_simple", O.make_params("IAN_simple", 1), dtype=torch.float64)
    worst = compare(replay(_TwinFacade(tw), fx), fx, tol=1e-6, max_off_by_one_frac=1e-3)
    assert worst["Z"] < 1e-6


# ---- the fixtures are what the reference produces today (build container only) -------------------------------
@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="reference checkout only exists in the build container")
def test_fixtures_regenerate_from_the_reference(tmp_path):
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_ref_golden", os.path.join(GOLD, "make_ref_golden.py"))
    mk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mk)
    from oracle.refexec.install import reference_modules
    with reference_modules() as ref:
        mk.gen_made_masks(ref, str(tmp_path / "m.npz"))
        mk.gen_layers(ref, str(tmp_path / "l.npz"))
        mk.gen_inference(ref, "IAN_simple", str(tmp_path / "s.npz"), str(tmp_path))
        mk.gen_session(ref, "IAN_simple", str(tmp_path / "e.npz"), str(tmp_path))
    for new, old in (("m.npz", "ref_made_masks.npz"), ("l.npz", "ref_layers.npz"), ("s.npz", "ref_IAN_simple.npz"), ("e.npz", "ref_session_IAN_simple.npz")):
        a, b = np.load(str(tmp_path / new)), load(old)
        assert sorted(a.files) == sorted(b.files)
        for k in a.files:   # float64 sums may be re-associated between runs (threaded torch-CPU convolutions)
            assert np.array_equal(a[k], b[k]) or rel(a[k], b[k]) < 1e-11, (old, k)
