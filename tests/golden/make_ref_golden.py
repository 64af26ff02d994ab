"""Generates tests/golden/ref_*.npz by EXECUTING the reference's own files.

    python tests/golden/make_ref_golden.py            (build container only: needs /root/reference)

/root/reference/{layers,mask_generator,IAN,IAN_simple,API,GANcheckpoints,train_IAN,sample_IAN}.py are imported
unmodified on top of the evaluating Theano/Lasagne stand-in in oracle/refexec/ (float64 torch-CPU arithmetic,
autograd for T.grad).  What their lines compute on seeded synthetic parameters and inputs is written here as
small fixtures; tests/test_reference_pinned.py (CPU) compares the oracle restatement with them and
tests/test_gpu_reference_pinned.py compares the HIP path with them.  Nothing in this file or in oracle/refexec/
is importable on the GPU box; only the .npz files travel.

Assumptions that remain [recalled] (oracle/refexec/minilasagne.py header): third-party primitive conventions
(cuDNN / Lasagne conv, transposed conv, dilated conv, batch_norm, nonlinearities, adam) and RandomStreams seeding.
Random normal draws (GaussianSampleLayer, layers.py:433) are recorded and stored, not reproduced bit-for-bit.

Fixtures:
  ref_made_masks.npz   MaskGenerator/MADE masks after API.py:33-36's reset("Once") and train_IAN.py:404's shuffle("Once")
  ref_layers.npz       layers.py building blocks on small shapes (MDCL, MDBLOCK, beta_layer, DeconvLayer,
                       MinibatchLayer, GaussianSampleLayer, IAFLayer, MADE)
  ref_IAN_simple.npz   API.IAN(IAN_simple.py): encode_images / sample_at / imgrad / imgradRGB, dnn=True and dnn=False
  ref_IAN.npz          API.IAN(IAN.py via a 3-line get_model(dnn=) shim) + sample_IAN.py's four functions
  ref_session_IAN_simple.npz   one NPE.py editing session as the sequence of facade calls its callbacks make (infer, 6 brush events in photo
                       mode with two colours / sizes, 3 scroll events, Reset), the numpy / scipy lines between them restated from NPE.py
  ref_train_IAN.npz    train_IAN.make_training_functions: update_gen / update_discrim metrics, every gradient
                       (recovered from the Adam first moments), parameter values after the two updates
"""
import logging
import os
import sys
import tempfile
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from neural_photo_editor_amd import synthetic as S  # noqa: E402  (pure numpy generators)
from oracle.refexec import minitheano as MT  # noqa: E402
from oracle.refexec.install import reference_modules  # noqa: E402

PATCHES = [(26, 26, 30, 30), (0, 0, 64, 64), (0, 0, 1, 1), (63, 63, 64, 64), (10, 20, 30, 40), (60, 0, 64, 9)]
SHIM = '''# get_model(dnn=) adapter: API.py:21 passes dnn=, IAN.get_model (IAN.py:67) only takes interp (README.md:71)
from IAN import *          # noqa  (the reference's IAN.py, unmodified)
from IAN import get_model as _reference_get_model
def get_model(dnn=True, interp=False):
    return _reference_get_model(interp=interp)
'''


def red_rgb():
    rgb = np.full((1, 3, 64, 64), -1.0, np.float32)
    rgb[:, 0] = 1.0
    return rgb


def sample_of(a, limit=4096):
    """A strided sample of a large activation plus its sum and abs-sum (keeps the fixtures small)."""
    flat = np.asarray(a, np.float64).ravel()
    stride = max(1, -(-flat.size // limit))
    if stride > 1 and stride % 2 == 0:
        stride += 1  # odd stride: does not alias with power-of-two extents
    return flat[::stride].copy(), np.array([stride, flat.sum(), np.abs(flat).sum()], np.float64)


def workdir_for(arch, P, tmp):
    d = os.path.join(tmp, arch)
    os.makedirs(d, exist_ok=True)
    cfg = os.path.join(d, arch + ".py")
    if arch == "IAN_simple":
        os.symlink(os.path.join("/root/reference", "IAN_simple.py"), cfg)
    else:
        with open(cfg, "w") as f:
            f.write(SHIM)
    np.savez(os.path.join(d, arch + ".npz"), **P)  # loaded by the reference's GANcheckpoints.load_weights (API.py:30)
    return cfg


class _Capture(logging.Handler):
    def __init__(self):
        logging.Handler.__init__(self)
        self.records = []

    def emit(self, record):
        self.records.append(record.getMessage())


def named_layers(ref, model):
    import lasagne
    out = {}
    for l in lasagne.layers.get_all_layers(model["l_out"]):
        if l.name:
            out[l.name] = l
    return out


# ------------------------------------------------------------------------------------------------------------
def gen_made_masks(ref, out):
    """The masks the reference's MaskGenerator/MADE produce (mask_generator.py:15-103, layers.py:653-853)."""
    import lasagne
    z = lasagne.layers.InputLayer((None, 100))
    res = {}
    for which, how in (("api", "reset"), ("train", "shuffle")):
        made = ref.layers.MADE(z, [100], "l_IAF_mu")
        if how == "reset":
            made.reset("Once")      # API.py:35, sample_IAN.py:165
        else:
            made.shuffle("Once")    # train_IAN.py:404
        masks = [l.weights_mask.get_value() for l in made.layers]   # input, output_W, output_D
        ordering = made.mask_generator.ordering.get_value()
        res[which] = (ordering, masks)
        if how == "shuffle":
            made.shuffle("Once")  # a second "Once" is a no-op (layers.py:832-837); after reset() it would NOT be
            assert all(np.array_equal(a, l.weights_mask.get_value()) for a, l in zip(masks, made.layers))
    (o1, m1), (o2, m2) = res["api"], res["train"]
    assert np.array_equal(o1, o2) and all(np.array_equal(a, b) for a, b in zip(m1, m2)), \
        "reset('Once') and shuffle('Once') must give the same masks"
    M = np.stack(m1)
    assert set(np.unique(M).tolist()) <= {0.0, 1.0}
    child = made.mask_generator._rng.state_updates[0][0].child_seed
    np.savez_compressed(out, ordering=o1.astype(np.int64), packed=np.packbits(M.astype(np.uint8)),
                        counts=M.reshape(3, -1).sum(1).astype(np.int64), child_seed=np.int64(child))
    print("made masks: counts", M.reshape(3, -1).sum(1), "ordering[:8]", o1[:8])


# ------------------------------------------------------------------------------------------------------------
def gen_layers(ref, out):
    """layers.py building blocks evaluated on small tensors."""
    import lasagne
    import theano
    import theano.tensor as T
    from lasagne.nonlinearities import LeakyRectify as lrelu
    rs = np.random.RandomState(11)
    lasagne.random.set_rng(np.random.RandomState(5))   # initialisers + the seed GaussianSampleLayer draws (layers.py:421)
    fx = {}

    def run(layer, feeds, **kw):
        keys = list(feeds)
        syms = [T.TensorType("float32", [False] * feeds[k].ndim)() for k in keys]
        expr = lasagne.layers.get_output(layer, dict(zip(keys, syms)), **kw)
        return theano.function(syms, expr)(*[feeds[k] for k in keys])

    def set_params(layer, prefix):
        """Randomises every shared variable under ``layer`` (seeded) and records it under its Theano name."""
        for p in lasagne.layers.get_all_params(layer):
            v = p.get_value()
            if p.name.endswith("weights_mask"):
                continue
            if p.name.endswith("inv_std") or p.name.endswith("gamma"):
                nv = rs.uniform(0.5, 1.5, v.shape)
            elif "_coeff_" in p.name:
                nv = v * rs.uniform(0.7, 1.3, v.shape)
            else:
                nv = rs.normal(0, 0.3, v.shape)
            p.set_value(nv.astype(np.float32))
            fx["%s/%s" % (prefix, p.name)] = p.get_value()

    # --- MDCL (layers.py:207-258), every scale set the configs use ---------------------------------------
    for tag, cin, cout, hw, scales in (("mdcl_02", 6, 5, 9, [0, 2]), ("mdcl_023", 4, 4, 11, [0, 2, 3]),
                                       ("mdcl_234", 5, 2, 12, [2, 3, 4])):
        l_in = lasagne.layers.InputLayer((None, cin, hw, hw))
        m = ref.layers.MDCL(l_in, cout, scales, "m")
        set_params(m, tag)
        x = rs.normal(0, 1, (2, cin, hw, hw)).astype(np.float32)
        fx[tag + "/x"], fx[tag + "/y"] = x, run(m, {l_in: x})
    # --- MDBLOCK (layers.py:411-416): deterministic and batch-statistics ---------------------------------
    l_in = lasagne.layers.InputLayer((None, 6, 8, 8))
    blk = ref.layers.MDBLOCK(l_in, 6, [0, 2, 3], "blk", lrelu(0.2))
    set_params(blk, "mdblock")
    x = rs.normal(0, 1, (3, 6, 8, 8)).astype(np.float32)
    fx["mdblock/x"] = x
    fx["mdblock/y_det"] = run(blk, {l_in: x}, deterministic=True)
    fx["mdblock/y_train"] = run(blk, {l_in: x}, deterministic=False)
    # --- beta_layer (layers.py:397-408) ---------------------------------------------------------------------
    la, lb = lasagne.layers.InputLayer((None, 1, 5, 5)), lasagne.layers.InputLayer((None, 1, 5, 5))
    a, b = rs.uniform(0, 1, (2, 1, 5, 5)).astype(np.float32), rs.uniform(0, 1, (2, 1, 5, 5)).astype(np.float32)
    fx["beta/a"], fx["beta/b"] = a, b
    fx["beta/y"] = run(ref.layers.beta_layer(la, lb), {la: a, lb: b})
    # --- DeconvLayer (layers.py:436-483) vs TransposedConv2DLayer crop=1 + slice (IAN_simple.py:182-223) ----
    l_in = lasagne.layers.InputLayer((None, 5, 4, 4))
    dc = ref.layers.DeconvLayer(l_in, 3, [5, 5], stride=[2, 2], crop=(2, 2), W=lasagne.init.Normal(0.3), b=None,
                                nonlinearity=None, name="dc")
    set_params(dc, "deconv")
    x = rs.normal(0, 1, (2, 5, 4, 4)).astype(np.float32)
    fx["deconv/x"], fx["deconv/y"] = x, run(dc, {l_in: x})
    from lasagne.layers import SliceLayer as SL, TransposedConv2DLayer as TC2D
    tc = SL(SL(TC2D(l_in, 3, [5, 5], stride=[2, 2], crop=(1, 1), W=dc.W, b=None, nonlinearity=None),
               indices=slice(1, None), axis=2), indices=slice(1, None), axis=3)
    assert np.abs(run(tc, {l_in: x}) - fx["deconv/y"]).max() < 1e-12
    # --- MinibatchLayer (layers.py:486-524) -------------------------------------------------------------------
    l_in = lasagne.layers.InputLayer((None, 7))
    mb = ref.layers.MinibatchLayer(l_in, num_kernels=6, dim_per_kernel=5, name="mb")
    set_params(mb, "minibatch")
    x = rs.normal(0, 1, (5, 7)).astype(np.float32)
    fx["minibatch/x"], fx["minibatch/y"] = x, run(mb, {l_in: x})
    # --- GaussianSampleLayer (layers.py:419-433) + IAFLayer (:641-650) + MADE (:735-853) ------------------------
    lmu, lls = lasagne.layers.InputLayer((None, 100)), lasagne.layers.InputLayer((None, 100))
    gs = ref.layers.GaussianSampleLayer(lmu, lls, name="gs")
    mu, ls = rs.normal(0, 1, (3, 100)).astype(np.float32), rs.normal(0, 0.3, (3, 100)).astype(np.float32)
    fx["gauss/mu"], fx["gauss/ls"] = mu, ls
    fx["gauss/y_det"] = run(gs, {lmu: mu, lls: ls}, deterministic=True)
    syms = [T.matrix(), T.matrix()]
    fn = theano.function(syms, lasagne.layers.get_output(gs, {lmu: syms[0], lls: syms[1]}))
    fx["gauss/y"] = fn(mu, ls)
    fx["gauss/eps"] = fn.last_draws[0][1]
    lz = lasagne.layers.InputLayer((None, 100))
    m_mu, m_ls = ref.layers.MADE(lz, [100], "l_IAF_mu"), ref.layers.MADE(lz, [100], "l_IAF_ls")
    iaf = ref.layers.IAFLayer(lz, m_mu, m_ls, name="iaf")
    m_mu.reset("Once")
    m_ls.reset("Once")
    for mm in (m_mu, m_ls):
        for p in mm.get_params():
            if not p.name.endswith("weights_mask"):
                v = p.get_value()
                p.set_value(rs.normal(0, 0.2, v.shape).astype(np.float32))
                fx["iaf/" + p.name] = p.get_value()
    z = rs.normal(0, 1, (3, 100)).astype(np.float32)
    fx["iaf/z"] = z
    fx["iaf/made_mu"] = run(m_mu, {lz: z})
    fx["iaf/made_ls"] = run(m_ls, {lz: z})
    fx["iaf/y"] = run(iaf, {lz: z})
    fx["iaf/final_layer_of_z"] = run(m_mu.final_layer, {lz: z})   # the masked MLP itself, fed with z
    np.savez_compressed(out, **fx)
    print("layers:", len(fx), "arrays")


# ------------------------------------------------------------------------------------------------------------
def gen_inference(ref, arch, out, tmp):
    """API.IAN on seeded synthetic parameters (API.py:11-110)."""
    import lasagne
    import theano
    import theano.tensor as T
    P = S.make_params(arch, 1)
    if arch == "IAN":
        P = S.make_train_params(P)
    lasagne.random.set_rng(np.random.RandomState(6))
    cfg = workdir_for(arch, P, tmp)
    cap = _Capture()
    logging.getLogger().addHandler(cap)
    t0 = time.time()
    model = ref.API.IAN(cfg, True)
    logging.getLogger().removeHandler(cap)
    missing = sorted(set(r.split()[4] for r in cap.records if r.startswith("unable to load parameter")))
    # every parameter the reference asks for must exist under the synthetic generator's names (App. B.5)
    allowed = {"minibatch_discrim.theta", "minibatch_discrim.log_weight_scale", "minibatch_discrim.b", "discrimi.W"} \
        if arch == "IAN_simple" else set()
    assert set(missing) <= allowed, missing
    asked = set(p.name for p in lasagne.layers.get_all_params(model.model["l_out"]) + lasagne.layers.get_all_params(model.model["l_discrim"]))
    unused = sorted(k for k in P if k not in asked)
    assert not unused, "synthetic parameters the reference never loads: %s" % unused
    fx = {}
    x = S.make_images(2, 0)
    zs = S.make_latents(2, 2)
    fx["x"], fx["z_sample"] = x, zs
    fx["z"] = model.encode_images(x)
    fx["xhat"] = model.sample_at(fx["z"].astype(np.float32)).astype(np.float32)
    fx["x_sample"] = model.sample_at(zs).astype(np.float32)
    assert model.get_zdim() == 100
    for k, (c1, r1, c2, r2) in enumerate(PATCHES):
        fx["grad_rgb_%d" % k] = model.imgradRGB(c1, r1, c2, r2, red_rgb(), zs[:1])
        fx["grad_light_%d" % k] = model.imgrad(c1, r1, c2, r2, zs[:1])
    fx["patches"] = np.asarray(PATCHES, np.int64)
    # a 10-step brush trajectory with the reference's own update rule (NPE.py:199-209)
    Z = zs[:1].astype(np.float64).copy()
    c1, r1, c2, r2 = PATCHES[0]
    for _ in range(10):
        Z = Z - 0.05 * model.imgradRGB(c1, r1, c2, r2, red_rgb(), Z.astype(np.float32)) * (1 + (c2 - c1))
        Z = Z.astype(np.float32).astype(np.float64)  # NPE keeps Z in float32
    fx["z_after_10_brush_steps"] = Z
    if arch == "IAN_simple":
        m2 = ref.API.IAN(cfg, False)  # the Lasagne-only decoder branch (IAN_simple.py:182-223)
        d = np.abs(m2.sample_at(zs) - model.sample_at(zs)).max()
        assert d < 1e-12, d
        fx["zpre"] = fx["z"]  # l_Z is the GaussianSampleLayer itself
    else:
        with_fns, _, _ = ref.sample_IAN.make_training_functions(model.cfg, model.model)  # sample_IAN.py:44-101
        fx["zpre"] = with_fns["Zfn"](x)
        fx["z_iaf_of_sample"] = with_fns["Z_IAF_fn"](zs)
        assert np.abs(with_fns["Z_IAF_fn"](fx["zpre"].astype(np.float32)) - fx["z"]).max() < 1e-5
        fx["x_from_ziaf"] = with_fns["sample"](zs).astype(np.float32)
        assert np.abs(with_fns["sampleZ"](zs) - fx["x_sample"]).max() < 1e-6
    # per-layer activations for one latent, sampled
    layers = named_layers(ref, model.model)
    want = {}
    if arch == "IAN_simple":
        want = {"dec_fc2": layers["bnorm_dec_fc2_nonlin"], "dec_conv1": layers["bnorm_dc1_nonlin"],
                "dec_conv2": layers["bnorm_dc2_nonlin"], "dec_conv3": layers["bnorm_dc3_nonlin"]}
    else:
        want = {"dec_fc2": layers["l_dec_fc2"], "dec_conv1": layers["dec_conv1"], "dec_conv2a": layers["dec_conv2"].input_layer,
                "dec_conv2": layers["dec_conv2"], "dec_conv3a": layers["dec_conv3"].input_layer,
                "dec_conv3": layers["dec_conv3"], "dec_conv4a": layers["dec_conv4"].input_layer,
                "dec_conv4": layers["bnorm_dc4_nonlin"]}
    Zs = T.matrix()
    names = list(want)
    fn = theano.function([Zs], lasagne.layers.get_output([want[n] for n in names], {model.model["l_Z"]: Zs}, deterministic=True))
    for n, a in zip(names, fn(zs[:1])):
        fx["act_" + n], fx["actstat_" + n] = sample_of(a)
    # encoder features
    Xs = T.tensor4()
    fn = theano.function([Xs], lasagne.layers.get_output(model.model["l_introspect"], {model.model["l_in"]: Xs}, deterministic=True))
    for i, a in enumerate(fn(x[:1])):
        fx["act_enc_conv%d" % (i + 1)], fx["actstat_enc_conv%d" % (i + 1)] = sample_of(a)
    np.savez_compressed(out, **fx)
    print("%s: %.0f s, z std %.4f, xhat std %.4f, |grad_rgb| %.3e" % (arch, time.time() - t0, fx["z"].std(), fx["xhat"].std(),
                                                                      np.abs(fx["grad_rgb_0"]).max()))


# ------------------------------------------------------------------------------------------------------------
def run_train(ref, tmp, B, order=("gen", "discrim")):
    """train_IAN.make_training_functions (train_IAN.py:47-352) on the full IAN config, B images per update: update_gen on
    batch 0 then update_discrim on batch 1, as the loop of train_IAN.py:497-504 alternates them.
    -> dict with metrics, the recorded epsilon draws, every gradient, the parameters after both updates."""
    import lasagne
    P = S.make_train_params(S.make_params("IAN", 1))
    lasagne.random.set_rng(np.random.RandomState(7))
    model = ref.IAN.get_model(interp=False)
    cfg = dict(ref.IAN.cfg)
    cfg["batch_size"] = B
    params = lasagne.layers.get_all_params(model["l_out"]) + lasagne.layers.get_all_params(model["l_discrim"])
    params = [p for i, p in enumerate(params) if p not in params[:i] and not p.name.endswith("weights_mask")]
    npz = os.path.join(tmp, "train_params.npz")
    np.savez(npz, **P)
    ref.GANcheckpoints.load_weights(npz, params)
    tfuncs, tvars, model = ref.train_IAN.make_training_functions(cfg, model)
    model["l_IAF_mu"].shuffle("Once")   # train_IAN.py:404-405
    model["l_IAF_ls"].shuffle("Once")
    X = S.make_images(2 * B, 21)
    Zr = np.random.RandomState(22).randn(2 * B, 100).astype(np.float32)
    tvars["X_shared"].set_value(X)                                        # train_IAN.py:477-484
    tvars["Z_shared"].set_value(Zr)
    tvars["p1"].set_value(np.asarray([[1, 0, 0]] * len(X), dtype=np.int32))
    tvars["p2"].set_value(np.asarray([[0, 1, 0]] * len(X), dtype=np.int32))
    tvars["p3"].set_value(np.asarray([[0, 0, 1]] * len(X), dtype=np.int32))
    R = {"X": X, "Z": Zr, "P": P, "lr": float(tvars["learning_rate"].get_value()), "grads": {}, "metrics": {}, "eps": {},
         "metric_names": {"gen": list(tvars["gd"]), "discrim": list(tvars["dd"])}}
    by_name = {p.name: p for p in params}

    def moments(fn):
        """{param name: m shared} for the Adam instances inside a compiled update function."""
        return {sv.adam_moment_of[0].name: sv for sv in fn.updates if getattr(sv, "adam_moment_of", (0, 0))[1] == "m"}

    def counters(fn):
        return [sv for sv in fn.updates if getattr(sv, "adam_step_counter", False)]

    prev_m = {}
    b1 = cfg["beta1"]
    for bi, tag in enumerate(order):                                                      # itr 0, 1 (:497)
        fn = tfuncs["update_" + tag]
        vals = fn(bi)
        R["metrics"][tag] = np.asarray([float(v) for v in vals], np.float64)
        assert len(fn.last_draws) == 1
        R["eps"][tag] = fn.last_draws[0][1]
        R["grads"][tag] = {}
        for name, sv in moments(fn).items():
            m_new = sv.get_value_f64().astype(np.float64)
            R["grads"][tag][name] = (m_new - b1 * prev_m.get(name, 0.0)) / (1 - b1)   # m_t = b1*m_prev + (1-b1)*g
            prev_m[name] = m_new
    # the Z group is stepped by both functions through ONE Adam instance (train_IAN.py:266-276)
    zc = [c for c in counters(tfuncs["update_gen"]) if c in counters(tfuncs["update_discrim"])]
    assert len(zc) == 1 and float(zc[0].get_value()) == float(len(order))
    if len(order) == 2:
        assert all(float(c.get_value()) == 1.0 for c in counters(tfuncs["update_gen"]) + counters(tfuncs["update_discrim"]) if c is not zc[0])
    R["after"] = {n: p.get_value_f64().astype(np.float64) for n, p in by_name.items()}
    return R


def gen_train(ref, out, tmp, B=4):
    t0 = time.time()
    R = run_train(ref, tmp, B)                      # update_gen(0) then update_discrim(1): the alternation of :497-504
    D = run_train(ref, tmp, B, order=("discrim",))  # update_discrim(0) from the SAME initial parameters ('discrim0')
    # the SAME reference graph evaluated in float32 (same seeds -> same epsilon): how far float32 arithmetic alone moves
    # each gradient of a step taken from the initial parameters.  Stored per tensor as 'noise32': the conditioning a
    # float32 implementation is judged against.  (The second step of a sequence also inherits the trajectory split of
    # Adam's sign-like first step, so its noise is not a property of the step: 'discrim' carries none.)
    MT.set_float_precision(32)
    try:
        R32 = run_train(ref, tmp, B, order=("gen",))
        D32 = run_train(ref, tmp, B, order=("discrim",))
    finally:
        MT.set_float_precision(64)
    for src, dst in ((D, R), (D32, R32)):
        for k in ("grads", "metrics", "eps"):
            dst[k]["discrim0"] = src[k]["discrim"]
        dst["metric_names"]["discrim0"] = src["metric_names"]["discrim"]
    fx = {"X": R["X"], "Z": R["Z"], "batch": np.int64(B), "lr": np.float64(R["lr"])}
    P = R["P"]
    rel = lambda a, b: float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))
    for tag in ("gen", "discrim", "discrim0"):
        fx[tag + "/metrics"], fx[tag + "/metric_names"], fx[tag + "/eps"] = R["metrics"][tag], np.asarray(R["metric_names"][tag]), R["eps"][tag]
        fx[tag + "/params"] = np.asarray(sorted(R["grads"][tag]))
        if tag != "discrim":
            assert np.array_equal(R["eps"][tag], R32["eps"][tag])
            fx[tag + "/metrics32"] = R32["metrics"][tag]
        for name, g in R["grads"][tag].items():
            if g.size <= 4096:
                fx["%s/grad/%s" % (tag, name)] = g
            else:
                fx["%s/grad_sample/%s" % (tag, name)], st = sample_of(g, 1024)
                fx["%s/grad_stat/%s" % (tag, name)] = np.array([st[0], st[1], st[2], np.sqrt((g * g).sum())])
            if tag != "discrim":
                fx["%s/noise32/%s" % (tag, name)] = np.float64(rel(R32["grads"][tag][name], g))
    trained = sorted(set(R["grads"]["gen"]) | set(R["grads"]["discrim"]))
    for name in trained:
        v = R["after"][name]
        if v.size <= 4096:
            fx["after/" + name] = v
        else:
            fx["after_sample/" + name], _ = sample_of(v, 1024)
    untouched = sorted(n for n in R["after"] if n not in trained)
    fx["untrained"] = np.asarray(untouched)
    for n in untouched:
        if not (n.endswith(".mean") or n.endswith(".inv_std")):
            assert np.array_equal(R["after"][n].astype(np.float32), P[n]), n
    np.savez_compressed(out, **fx)
    print("train: %.0f s  gen %s  discrim %s" % (time.time() - t0, np.round(fx["gen/metrics"], 5), np.round(fx["discrim/metrics"], 5)))
    for tag in ("gen", "discrim0"):
        nz = np.array([float(fx["%s/noise32/%s" % (tag, n)]) for n in R["grads"][tag]])
        print("  %s: %d params; float32 evaluation of the reference graph moves the gradients by median %.2e, max %.2e (rel. max-norm)"
              % (tag, len(nz), np.median(nz), nz.max()))
    print("  never trained:", [n for n in untouched if "bnorm" not in n and "_bn" not in n][:12])


# ---- one editing session: the facade-call sequence of NPE.py's callbacks (no UI) ---------------------------------------------------
# NPE.py itself cannot be imported (Python 2 print statements, Tkinter widgets built at import time: NPE.py:14-35, 60-130), so the few
# numpy / scipy lines BETWEEN its model calls are restated here, dtype for dtype, each citing the line it follows; every model call is
# the reference's own API.IAN method on the evaluating stand-in.  The stand-in computes in float64 where Theano (floatX=float32,
# README.md:19-21) returns float32: results of model calls are cast to float32 before NPE's arithmetic touches them.
SESSION_EVENTS = (
    # (kind, payload): what the user does, in order.  Coordinates are canvas pixels (the output canvas is 4 x 64 wide, NPE.py:148)
    ("infer", None),                         # NPE.py:239-279 on the synthetic stand-in for CelebAValid[val]
    ("color", (230, 40, 30)),                # getColor NPE.py:353-359
    ("size", 12),                            # d.get() NPE.py:100-101 -> brush_width 4
    ("paint", (114, 113)), ("paint", (118, 114)), ("paint", (121, 118)), ("paint", (125, 121)),      # B1-Motion, NPE.py:192-235
    ("color", (20, 60, 220)),
    ("size", 28),                            # brush_width 8
    ("paint", (40, 162)), ("paint", (47, 165)),
    ("scroll", +1), ("scroll", +1), ("scroll", -1),     # NPE.py:305-316, at the rectangle the brush was left at
    ("reset", None),                         # NPE.py:330-340
)


def brush_rect(ex, ey, dsize):
    """move_mouse NPE.py:142-156 + the // 4 of NPE.py:202: (x1, y1, x2, y2) in 64-pixel space."""
    x, y = ex // 4, ey // 4
    bw = (dsize // 4) + 1
    xmin = max(min(x - bw // 2, 64 - bw), 0)
    ymin = max(min(y - bw // 2, 64 - bw), 0)
    return xmin, ymin, xmin + bw, ymin + bw


def gen_session(ref, arch, out, tmp):
    """-> ref_session_<arch>.npz: inputs (GIM, the event list) and, after every event, the latent and what the canvas would show."""
    import lasagne
    import scipy.ndimage
    to_tanh = lambda a: 2.0 * (a / 255.0) - 1.0           # NPE.py:37-38
    from_tanh = lambda a: 255.0 * (a + 1) / 2.0           # NPE.py:40-41
    f32 = lambda a: np.asarray(a).astype(np.float32)      # a Theano function's float32 return value
    P = S.make_params(arch, 1)
    if arch == "IAN":
        P = S.make_train_params(P)
    lasagne.random.set_rng(np.random.RandomState(6))
    model = ref.API.IAN(workdir_for(arch, P, os.path.join(tmp, "session")), True)
    GIM = np.uint8((S.make_images(1, seed=9)[0] + 1.0) * 127.5)         # stands in for np.load('CelebAValid.npz')['arr_0'][val]
    fx = {"GIM": GIM}
    Z = np.zeros((10, 10), np.float32)
    myRGB = np.zeros((1, 3, 64, 64), dtype=np.float32)                   # NPE.py:87
    dsize, rect = 12, (0, 0, 4, 4)
    IM = RECON = ERROR = None
    kinds, recs = [], []
    for kind, arg in SESSION_EVENTS:
        shown = None
        if kind in ("infer", "reset"):                                    # NPE.py:257-279 / 330-340: the same five lines
            IM = GIM
            s_ = f32(model.encode_images(np.asarray([to_tanh(IM)], dtype=np.float32)))
            Z = np.reshape(s_[0], np.shape(Z))
            RECON = np.uint8(from_tanh(f32(model.sample_at(np.float32([Z.flatten()])))[0]))
            ERROR = to_tanh(np.float32(IM)) - to_tanh(np.float32(RECON))
            shown = IM
            fx["%02d_RECON" % len(kinds)], fx["%02d_ERROR" % len(kinds)] = RECON, ERROR
        elif kind == "color":
            for i in range(3):
                myRGB[0, i, :, :] = arg[i]                                # NPE.py:359
        elif kind == "size":
            dsize = arg
        elif kind == "paint":                                             # NPE.py:199-231, photo mode (SAMPLE_FLAG = 0 after infer)
            rect = brush_rect(arg[0], arg[1], dsize)
            x1, y1, x2, y2 = rect
            temp = np.asarray(f32(model.imgradRGB(x1, y1, x2, y2, np.float32(to_tanh(myRGB)), np.float32([Z.flatten()])))[0])
            grad = temp.reshape((10, 10)) * (1 + (x2 - x1))
            Z -= 0.05 * grad
            DELTA = f32(model.sample_at(np.float32([Z.flatten()])))[0] - to_tanh(np.float32(RECON))
            MASK = scipy.ndimage.gaussian_filter(np.min([np.mean(np.abs(DELTA), axis=0), np.ones((64, 64))], axis=0), 0.7)
            D = MASK * DELTA + (1 - MASK) * ERROR
            IM = np.uint8(from_tanh(to_tanh(RECON) + D))
            shown = IM
            fx["%02d_MASK" % len(kinds)] = MASK
        elif kind == "scroll":                                            # NPE.py:310-316; event.delta = +-1
            x1, y1, x2, y2 = rect
            grad = np.reshape(f32(model.imgrad(x1, y1, x2, y2, np.float32([Z.flatten()])))[0], Z.shape) * (1 + (x2 - x1))
            Z += np.sign(arg) * 0.1 * grad
            shown = np.uint8(from_tanh(f32(model.sample_at(np.float32([Z.flatten()])))[0]))     # update_photo(None) NPE.py:110
        assert Z.dtype == np.float32
        k = len(kinds)
        kinds.append(kind)
        recs.append(list(rect) + [dsize] + [int(v) for v in myRGB[0, :, 0, 0]] + ([int(arg)] if kind == "scroll" else [0]))
        fx["%02d_Z" % k] = Z.copy()
        if shown is not None:
            fx["%02d_shown" % k] = np.asarray(shown).copy()
    fx["kinds"] = np.asarray(kinds)
    fx["state"] = np.asarray(recs, np.int64)      # per event: x1, y1, x2, y2, brush size d, brush colour r g b, scroll delta
    np.savez_compressed(out, **fx)
    print("session %s: %d events, |Z| after infer %.3f, after the last scroll %.3f" % (arch, len(kinds), np.abs(fx["00_Z"]).max(), np.abs(fx["%02d_Z" % (len(kinds) - 2)]).max()))


def main(which):
    logging.basicConfig(level=logging.ERROR)
    with reference_modules() as ref, tempfile.TemporaryDirectory() as tmp:
        if "masks" in which:
            gen_made_masks(ref, os.path.join(HERE, "ref_made_masks.npz"))
        if "layers" in which:
            gen_layers(ref, os.path.join(HERE, "ref_layers.npz"))
        for arch in ("IAN_simple", "IAN"):
            if arch in which:
                gen_inference(ref, arch, os.path.join(HERE, "ref_%s.npz" % arch), tmp)
        if "train" in which:
            gen_train(ref, os.path.join(HERE, "ref_train_IAN.npz"), tmp)
        if "session" in which:
            gen_session(ref, "IAN_simple", os.path.join(HERE, "ref_session_IAN_simple.npz"), tmp)


if __name__ == "__main__":
    main(sys.argv[1:] or ["masks", "layers", "IAN_simple", "IAN", "train", "session"])
