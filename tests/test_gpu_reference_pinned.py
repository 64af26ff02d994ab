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


@pytest.mark.parametrize("arch", ARCHS)
def test_bf16x3_option_is_opt_in_and_inside_the_parity_bar(arch):
    """ian_set_option("tg_bf16x3", 1): tap-GEMMs on the bf16 matrix cores with split operands (three MFMAs per product,
    kernels_tapgemm.hip tapgemm_bf16x3_kernel) -- a labelled secondary of the bench line, never the default.  Held to the SAME
    reference-executed fixture and the same 1e-4 bar as the exact-fp32 path (layers.py:476-481, IAN_simple.py:141-170 are what
    these launches replace), with tg_bf16x3_min_m = 1 pushing every multi-image layer whose tile the kernel supports through it;
    and at the benchmarked batch against the exact-fp32 HIP path on the same inputs.
    The batch-1 chains stay exact fp32 under the option (tg_bf16x3_min_images = 2): the brush gradient API.py:59,64 is
    DISCONTINUOUS in the forward activations (leaky-ReLU / ReLU kinks), and a 1e-5 forward error flips a unit under the patch --
    measured 1.5e-3 / 3.0e-3 on patch 1 of this fixture with the batch-1 forward forced through the split-bf16 kernel
    (scripts/exp/bf16x3_brush_debug.py), while split-bf16 BACKWARD-data alone is smooth: held to the bar here."""
    from neural_photo_editor_amd import IAN
    fx = np.load(os.path.join(GOLD, "ref_%s.npz" % arch))
    m = IAN(os.path.join(CFG, arch + ".py"), True, params=S.make_params(arch, 1))
    ref32 = model_for(arch)
    try:
        B = 64 if arch == "IAN_simple" else 16
        xb = S.make_images(B, seed=100)
        before = m.reconstruct(xb)
        assert np.array_equal(before, ref32.reconstruct(xb))          # off by default: bitwise the fp32 path
        m.handle.set_option("tg_bf16x3", 1)
        m.handle.set_option("tg_bf16x3_min_m", 1)
        x, zs = fx["x"], fx["z_sample"]
        errs = {"z": rel(m.encode_images(x), fx["z"]), "xhat_of_z": rel(m.sample_at(fx["z"].astype(np.float32)), fx["xhat"]),
                "x_sample": rel(m.sample_at(zs), fx["x_sample"]), "recon": rel(m.reconstruct(x), fx["xhat"])}
        after = m.reconstruct(xb)
        errs["recon_batch%d_vs_fp32_hip" % B] = rel(after, before)
        assert not np.array_equal(after, before)                      # the option did switch kernels
        z = zs[:1]
        patches = fx["patches"].tolist()
        for c1, r1, c2, r2 in patches:                                # batch 1: untouched by the option, bit for bit
            assert np.array_equal(m.imgradRGB(c1, r1, c2, r2, red_rgb(), z), ref32.imgradRGB(c1, r1, c2, r2, red_rgb(), z))
        m.handle.set_option("tg_bf16x3_min_images", 1)                # ... unless asked: backward-data launches only
        m.handle.set_option("tg_bf16x3_fwd", 0)
        g = 0.0
        for k, (c1, r1, c2, r2) in enumerate(patches):
            g = max(g, rel(m.imgradRGB(c1, r1, c2, r2, red_rgb(), z), fx["grad_rgb_%d" % k]),
                    rel(m.imgrad(c1, r1, c2, r2, z), fx["grad_light_%d" % k]))
        errs["brush_grad_split_bf16_backward_data_only"] = g
        _note("bf16x3_max_rel_err_" + arch, errs)
        for k, v in errs.items():
            assert v < TOL, (k, v, errs)
    finally:
        m.close()


def test_tune_cache_keeps_the_split_bf16_choices_apart(tmp_path, monkeypatch):
    """IAN_TUNE_CACHE is shared by every handle of a process tree (the rocprofv3 passes of one profile replay the first run's
    choices).  A handle in the opt-in tg_bf16x3 mode tunes OTHER kernels on other tiles: its entries carry their own direction key,
    so it neither replays the exact-fp32 choices (fp32 kernels on every tile the split-bf16 kernel is not instantiated for: the
    bench line's labelled secondary once read 1.17 instead of 0.66 ms that way) nor overwrites them."""
    from neural_photo_editor_amd import IAN
    cache = tmp_path / "tune.txt"
    monkeypatch.setenv("IAN_TUNE_CACHE", str(cache))
    x = S.make_images(8, seed=3)
    P = S.make_params("IAN_simple", 1)
    m = IAN(os.path.join(CFG, "IAN_simple.py"), True, params=P)
    try:
        ref = m.reconstruct(x)
        m.handle.autotune(8, 1)
        first = cache.read_text().splitlines()
        assert first[0].startswith("ian-tune-cache") and all(ln.split()[1] == "fwd" for ln in first[1:]) and len(first) > 5
        assert rel(m.reconstruct(x), ref) < 1e-5                      # tuned split-K limits move the fp32 summation order, nothing else
        m.handle.set_option("tg_bf16x3", 1)
        m.handle.set_option("tg_bf16x3_min_m", 1)
        m.handle.autotune(8, 1)
        both = cache.read_text().splitlines()
        dirs = [ln.split()[1] for ln in both[1:]]
        assert dirs.count("fwd") == len(first) - 1 and dirs.count("fwd+bf16x3") == len(first) - 1
        assert [ln for ln in both[1:] if ln.split()[1] == "fwd"] == first[1:]      # the exact-fp32 choices were not touched
        assert rel(m.reconstruct(x), ref) < TOL
    finally:
        m.close()


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
    """update_gen / update_discrim as train_IAN.py:497-504 drives them, same X / Z / epsilon the reference run used.
    Metrics: 2e-5 against the reference-executed float64 values (measured 1.4e-6).
    Gradients of a step taken from the initial parameters are compared per tensor (relative max-norm error) against the
    float64 reference execution and JUDGED AGAINST THE FIXTURE'S 'noise32': how far the reference's own graph moves when the
    stand-in evaluates it in float32.  What the comparison can and cannot show (round 4, measured):
      * the composed gradient of this fixture is ill-conditioned for ANY float32 implementation (batch statistics over 4
        near-identical decoder outputs, leaky-ReLU and |a_b - a_b'| kinks): noise32 itself reaches 17 % on l_dec_fc2.W, and
        1-ulp perturbations of the layer outputs move single tensors by 1e-2..6e-2 with a 7x spread from seed to seed
        (scripts/exp/fp32_noise_conditioning.py -> profiles/r04_fp32_conditioning.json);
      * tests/test_gpu_decomposition.py evaluates the float64 graph AT THE HIP STEP'S OWN ACTIVATIONS on this very minibatch:
        every stage's local forward error <= 2.7e-6, every gradient tensor within 3.1e-5 (median 1.2e-6) -- the residual seen
        here is forward drift x conditioning, not backward arithmetic (profiles/r04_decomposition.json).  That test is the
        sharp per-tensor guard; this one holds the DISTRIBUTION of the composed error to the float32 noise of the reference's
        own graph: group median <= 2 x noise32's median, 90 % of the tensors <= 3 x their own noise32 + 1e-3, and no tensor
        above 3 x the LARGEST noise32 of its update (one draw of a heavy-tailed quantity cannot bound another per tensor).
    Round 3 (float32 one-pass batch variance) measured discrim0 median 7.5e-4 / gen worst 0.113; with the float64 batch
    statistics of round 4: discrim0 median 9.3e-6 (noise32: 1.5e-5), gen median 1.4e-3 (1.1e-3), worst 1.6e-2 (noise32: 0.17)."""
    import torch
    from neural_photo_editor_amd.trainer import Trainer
    fx = np.load(os.path.join(GOLD, "ref_train_IAN.npz"))
    B = int(fx["batch"])
    P = S.make_train_params(S.make_params("IAN", 1))
    tr = Trainer(os.path.join(CFG, "IAN.py"), P, batch=B)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    X, Z = fx["X"], fx["Z"]
    report = {}
    for tag, which, groups in (("gen", "gen", ("dec", "Z")), ("discrim0", "discrim", ("enc", "Z"))):
        xb, zb, eps = dev(X[:B]), dev(Z[:B]), dev(fx[tag + "/eps"])
        tr.forward(xb, zb, eps)
        m = tr.metrics()
        names = fx[tag + "/metric_names"].tolist()
        for n, b in zip(names, fx[tag + "/metrics"]):
            if n != "discrim_acc":                                              # argmax count over 12 decisions
                assert abs(m[n] - b) <= 2e-5 * max(1.0, abs(b)), (tag, n, m[n], b)
        tr.backward(which)
        tr._regularizers(which)
        errs, noise = {}, {}
        for gname in groups:
            for name, g in tr.grads_numpy(gname).items():
                errs[name] = _grad_err(fx, tag, name, g)
                noise[name] = float(fx["%s/noise32/%s" % (tag, name)])
        assert sorted(errs) == sorted(fx[tag + "/params"].tolist())           # the reference's parameter groups
        ev, nv = np.array([errs[k] for k in errs]), np.array([noise[k] for k in errs])
        over = sorted(((errs[k], noise[k], k) for k in errs if errs[k] > 3 * noise[k] + 1e-3), reverse=True)
        report[tag] = {"median": float(np.median(ev)), "max": float(ev.max()), "float32_eval_median": float(np.median(nv)),
                       "float32_eval_max": float(nv.max()), "tensors": len(errs), "above_3x_own_noise_plus_1e-3": over,
                       "worst": sorted(((errs[k], noise[k], k) for k in errs), reverse=True)[:6],
                       "all": sorted(((errs[k], noise[k], k) for k in errs), reverse=True)}
        _note("train_grad_rel_err", report)
        assert np.median(ev) <= 2 * np.median(nv), report[tag]["worst"]
        assert len(over) <= 0.1 * len(errs), over
        assert ev.max() <= 3 * nv.max(), report[tag]["worst"]
    # the alternation itself: update_gen(batch 0) then update_discrim(batch 1); the second step inherits the split of
    # Adam's sign-like first step (|step| ~ lr whatever |g|), so its metrics are held to 1e-2 and the parameters to
    # "much closer to the reference's end point than to the start"
    got = np.array(tr.update_gen(dev(X[:B]), dev(Z[:B]), dev(fx["gen/eps"])), np.float64)
    assert np.allclose(got, fx["gen/metrics"], rtol=2e-5, atol=2e-5), (got, fx["gen/metrics"])
    got = np.array(tr.update_discrim(dev(X[B:]), dev(Z[B:]), dev(fx["discrim/eps"])), np.float64)
    keep = [i for i, n in enumerate(fx["discrim/metric_names"].tolist()) if n != "discrim_acc"]
    _note("second_update_metric_rel_err", {n: float(abs(got[i] - fx["discrim/metrics"][i]) / max(1.0, abs(fx["discrim/metrics"][i])))
                                           for i, n in enumerate(fx["discrim/metric_names"].tolist())})
    # the second update starts from parameters that took one float32 Adam step: measured 4.7e-6 (round 4; round 3 allowed 1e-2)
    assert np.allclose(got[keep], fx["discrim/metrics"][keep], rtol=2e-4, atol=2e-4), (got, fx["discrim/metrics"])
    assert tr.groups["Z"].t == 2 and tr.groups["dec"].t == 1 and tr.groups["enc"].t == 1     # ONE Adam instance for Z (:266-276)
    # Parameters after the two updates, PER TENSOR against the reference's end point (round-4 verdict item 6; the round-4 bar
    # was one aggregate, err.mean() < 0.1 * moved.mean(), which a sign error on a minority of tensors would have passed).
    # Adam's first step is sign-like -- |step| = lr whatever |g| -- so an element lands 2 lr away from the reference exactly
    # when its gradient's sign differs, i.e. when |g| is below the implementation's error: mean |after - reference| of a tensor
    # is 2 lr x (fraction of elements on the other side).
    #   * decoder_params are stepped ONCE, from the initial parameters (update_gen): held to 2 lr x (the tensor's own float32
    #     noise in the reference's graph + 1e-3), with a floor of 2.5 elements per tensor (a 256-element coefficient vector whose
    #     smallest gradient flips is 1/256 = 4e-3 on its own; measured: exactly such single elements, 17 tensors of 82 over the
    #     unfloored bound, profiles/r05_reference_pinned_errors.json);
    #   * encoder_params and Z_params are touched by the SECOND update, whose forward already runs on parameters that differ
    #     from the reference's by the first step's flipped elements -- their gradients are a different draw, for which the fixture
    #     holds no noise figure (discrim0 is the discriminator step from the INITIAL parameters): held to <= 10 % of the elements
    #     on the other side (0.2 lr; measured worst 4.2 % on the 256-element bnorm2.gamma).
    # A sign error on one whole tensor puts ALL its elements 2 lr away: 10x to 500x above either bar.
    after = tr.params_numpy()
    moved, per = [], {}
    for key in fx.files:
        if key.startswith("after/"):
            n = key[6:]
            ref = np.asarray(fx[key], np.float64)
            moved.append(np.abs(ref - np.asarray(P[n], np.float64)).ravel())
            e = float(np.abs(after[n] - ref).mean())
            if tr.where[n] == "dec":
                nz = float(fx["gen/noise32/" + n])
                bound = 2 * tr.lr * max(nz + 1e-3, 2.5 / ref.size)
            else:
                nz = max([float(fx["%s/noise32/%s" % (tag, n)]) for tag in ("gen", "discrim0") if "%s/noise32/%s" % (tag, n) in fx.files] or [0.0])
                bound = 0.2 * tr.lr
            per[n] = (e, bound, nz, tr.where[n], int(ref.size))
    moved = np.concatenate(moved)
    bad = sorted(((e / b, e, b, nz, g, n) for n, (e, b, nz, g, sz) in per.items() if e > b), reverse=True)
    _note("post_update_param_err", {"lr": tr.lr, "tensors": len(per), "over_bound": bad,
                                    "fraction_of_elements_on_the_other_side": sorted(((e / (2 * tr.lr), g, sz, n) for n, (e, b, nz, g, sz) in per.items()), reverse=True)[:12],
                                    "worst_ratio_to_bound": sorted(((e / b, e, nz, g, n) for n, (e, b, nz, g, sz) in per.items()), reverse=True)[:8]})
    assert moved.mean() > 0.5 * tr.lr, moved.mean()
    assert len(per) >= 60 and not bad, bad[:6]
    untouched = [n for n in fx["untrained"].tolist() if n.startswith("l_IAF")]
    frozen = tr.state_dict()
    assert len(untouched) == 12 and all(np.array_equal(frozen[n], P[n]) for n in untouched)   # MADE is never trained
