"""GPU parity against REFERENCE-EXECUTED fixtures (tests/golden/ref_*.npz): what the reference's own layers.py /
IAN*.py / API.py / sample_IAN.py / train_IAN.py computed (float64, on the evaluating Theano/Lasagne stand-in of
oracle/refexec, see tests/golden/make_ref_golden.py) versus the HIP path through the C ABI.  No oracle code sits
between the two sides here: the only things imported besides the product are numpy and the seeded input generators.
Tolerances: 1e-4 relative (max-abs-error / max-abs-reference) on float32 activations (north star); gradients as
stated per test; MADE masks bit-exact."""
import json
import os

import numpy as np
import pytest

from neural_photo_editor_amd import synthetic as S

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CFG = os.path.join(ROOT, "neural_photo_editor_amd", "configs")
GOLD = os.path.join(ROOT, "tests", "golden")
TOL = 1e-4
ARCHS = ("IAN_simple", "IAN")


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


def red_rgb():
    rgb = np.full((1, 3, 64, 64), -1.0, np.float32)
    rgb[:, 0] = 1.0
    return rgb


_cache = {}


def model_for(arch):
    if arch not in _cache:
        from neural_photo_editor_amd import IAN
        _cache[arch] = IAN(os.path.join(CFG, arch + ".py"), True, params=S.make_params(arch, 1))
    return _cache[arch]


def test_made_masks_bit_exact_vs_mask_generator():
    m = model_for("IAN")
    g = np.load(os.path.join(GOLD, "ref_made_masks.npz"))
    M = np.unpackbits(g["packed"])[:30000].reshape(3, 100, 100)
    for mine, theirs in zip(m.made_masks, M):
        assert np.array_equal(np.asarray(mine).astype(np.uint8), theirs)


@pytest.mark.parametrize("arch", ARCHS)
def test_encode_decode_vs_reference_api(arch):
    m = model_for(arch)
    fx = np.load(os.path.join(GOLD, "ref_%s.npz" % arch))
    x, zs = fx["x"], fx["z_sample"]
    assert rel(m.Zfn(x), fx["zpre"]) < TOL
    assert rel(m.encode_images(x), fx["z"]) < TOL                                   # API.py:78-90
    assert rel(m.sample_at(fx["z"].astype(np.float32)), fx["xhat"]) < TOL           # API.py:98-110
    assert rel(m.sample_at(zs), fx["x_sample"]) < TOL
    assert rel(m.reconstruct(x), fx["xhat"]) < TOL
    if arch == "IAN":                                                               # sample_IAN.py:86-94
        assert rel(m.Z_IAF_fn(zs), fx["z_iaf_of_sample"]) < TOL
        assert rel(m.sample(zs), fx["x_from_ziaf"]) < TOL


@pytest.mark.parametrize("arch", ARCHS)
def test_layer_activations_vs_reference_graph(arch):
    m = model_for(arch)
    fx = np.load(os.path.join(GOLD, "ref_%s.npz" % arch))
    m.encode_images(fx["x"][:1])
    for i in range(4):
        k = "enc_conv%d" % (i + 1)
        a = m.activation(k, 1)
        stride = int(fx["actstat_" + k][0])
        assert rel(a.ravel()[::stride], fx["act_" + k]) < TOL, k
    m.sample_at(fx["z_sample"][:1])
    rename = {"dec_fc2": "l_dec_fc2", "dec_conv2a": "dec_conv2a2", "dec_conv3a": "dec_conv3a2", "dec_conv4a": "dec_conv4a2"}
    checked = 0
    for key in fx.files:
        if not key.startswith("act_dec"):
            continue
        name = key[4:]
        nm = rename.get(name, name)
        if nm in m.lowered.slot_names:
            a = m.activation(nm, 1)
            stride = int(fx["actstat_" + name][0])
            assert rel(a.ravel()[::stride], fx[key]) < TOL, name
            checked += 1
    assert checked >= (4 if arch == "IAN_simple" else 7)


@pytest.mark.parametrize("arch", ARCHS)
def test_brush_gradients_vs_reference_api(arch):
    """API.py:59,64 through T.grad of the reference graph; bar = 1e-4 like the activations."""
    m = model_for(arch)
    fx = np.load(os.path.join(GOLD, "ref_%s.npz" % arch))
    z = fx["z_sample"][:1]
    worst = 0.0
    for k, (c1, r1, c2, r2) in enumerate(fx["patches"].tolist()):
        e1 = rel(m.imgradRGB(c1, r1, c2, r2, red_rgb(), z), fx["grad_rgb_%d" % k])
        e2 = rel(m.imgrad(c1, r1, c2, r2, z), fx["grad_light_%d" % k])
        worst = max(worst, e1, e2)
        assert e1 < TOL and e2 < TOL, (k, e1, e2)
    Z = z.copy()
    c1, r1, c2, r2 = fx["patches"][0].tolist()
    for _ in range(10):                                                             # NPE.py:199-209
        Z = (Z - 0.05 * m.imgradRGB(c1, r1, c2, r2, red_rgb(), Z) * (1 + (c2 - c1))).astype(np.float32)
    assert rel(Z, fx["z_after_10_brush_steps"]) < TOL
    _note("brush_grad_max_rel_err_" + arch, worst)


def test_made_iaf_kernel_vs_reference_layers():
    """ian_k_made_iaf on the small-layer fixture: IAFLayer(z, MADE, MADE) as layers.py wires it."""
    import torch
    from neural_photo_editor_amd import lib as L
    from neural_photo_editor_amd.trainer import K
    k = K(L.load_train_library())
    fx = np.load(os.path.join(GOLD, "ref_layers.npz"))
    g = np.load(os.path.join(GOLD, "ref_made_masks.npz"))
    M = np.unpackbits(g["packed"])[:30000].reshape(3, 100, 100).astype(np.float32)
    Ws, bs = [], []
    for made in ("l_IAF_mu", "l_IAF_ls"):
        for j, part in enumerate(("_input", "_output_W", "_output_D")):
            Ws.append(fx["iaf/%s%s.W" % (made, part)] * M[j])
            bs.append(fx["iaf/%s%s.b" % (made, part)])
    c = lambda v: torch.from_numpy(np.ascontiguousarray(v, np.float32)).cuda()
    z = fx["iaf/z"]
    zd = c(np.pad(z, ((0, 0), (0, 28))))
    out = torch.zeros_like(zd)
    k.made_iaf(zd, out, c(np.stack(Ws)), c(np.stack(bs)), z.shape[0], 100, 128)
    assert rel(out.cpu().numpy()[:, :100], fx["iaf/y"]) < TOL


# ---- training step ------------------------------------------------------------------------------------------------
_notes = {}


def _note(key, value):
    _notes[key] = value
    out = os.path.join(ROOT, "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "reference_pinned_errors.json"), "w") as f:
            json.dump(_notes, f, indent=1, sort_keys=True)
    except OSError:
        pass


def _grad_err(fx, tag, name, g):
    g = np.asarray(g, np.float64)
    if "%s/grad/%s" % (tag, name) in fx.files:
        return rel(g, fx["%s/grad/%s" % (tag, name)])
    st = fx["%s/grad_stat/%s" % (tag, name)]
    e1 = rel(g.ravel()[::int(st[0])], fx["%s/grad_sample/%s" % (tag, name)])
    e2 = abs(np.sqrt((g * g).sum()) - st[3]) / (st[3] + 1e-30)
    return max(e1, e2)


def test_training_functions_vs_reference_train_IAN():
    """update_gen then update_discrim exactly as train_IAN.py:497-504 drives them, same X / Z / epsilon the reference
    run used.  Metrics to 2e-4; gradients: per-tensor relative error (max-abs / max-abs), median and maximum bounded
    and recorded (float32 chain of ~25 layers with batch statistics over 4 images against a float64 reference)."""
    import torch
    from neural_photo_editor_amd.trainer import Trainer
    fx = np.load(os.path.join(GOLD, "ref_train_IAN.npz"))
    B = int(fx["batch"])
    P = S.make_train_params(S.make_params("IAN", 1))
    tr = Trainer(os.path.join(CFG, "IAN.py"), P, batch=B)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    X, Z = fx["X"], fx["Z"]
    report = {}
    for tag, lo, groups, fn in (("gen", 0, ("dec", "Z"), tr.update_gen), ("discrim", B, ("enc", "Z"), tr.update_discrim)):
        xb, zb, eps = dev(X[lo:lo + B]), dev(Z[lo:lo + B]), dev(fx[tag + "/eps"])
        # gradients of the composed step, before Adam consumes them
        tr.forward(xb, zb, eps)
        tr.backward(tag)
        tr._regularizers(tag)
        errs = {}
        for gname in groups:
            for name, g in tr.grads_numpy(gname).items():
                errs[name] = _grad_err(fx, tag, name, g)
        assert sorted(errs) == sorted(fx[tag + "/params"].tolist())           # the reference's parameter groups
        vals = np.array(list(errs.values()))
        report[tag] = {"median": float(np.median(vals)), "max": float(vals.max()),
                       "worst": sorted(errs.items(), key=lambda kv: -kv[1])[:5]}
        assert np.median(vals) < 1e-3, report[tag]
        assert vals.max() < 3e-2, report[tag]
        got = np.array(fn(xb, zb, eps), np.float64)                            # the update itself (recomputes the step)
        ref = fx[tag + "/metrics"]
        names = fx[tag + "/metric_names"].tolist()
        for n, a, b in zip(names, got, ref):
            if n == "discrim_acc":
                continue                                                        # argmax count over 12 decisions
            assert abs(a - b) <= 2e-4 * max(1.0, abs(b)), (tag, n, a, b)
    _note("train_grad_rel_err", report)
    # parameters after the two updates: Adam's first steps are sign-like (|step| ~ lr), so compare the MOVE
    after = tr.params_numpy()
    moved, err = [], []
    for key in fx.files:
        if key.startswith("after/"):
            n = key[6:]
            moved.append(np.abs(fx[key] - np.asarray(P[n], np.float64)).ravel())
            err.append(np.abs(after[n] - fx[key]).ravel())
    moved, err = np.concatenate(moved), np.concatenate(err)
    assert moved.mean() > 0.5 * tr.lr and err.mean() < 0.1 * moved.mean(), (moved.mean(), err.mean())
    untouched = [n for n in fx["untrained"].tolist() if n.startswith("l_IAF")]
    assert len(untouched) == 12 and all(np.array_equal(after[n], P[n]) for n in untouched)   # MADE is never trained
