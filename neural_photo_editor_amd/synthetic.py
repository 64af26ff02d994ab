"""Seeded synthetic parameters and inputs: the reference's trained weights (IAN_simple.npz,
IANv1.npz: git-LFS pointers) and CelebAValid.npz are absent, so the benchmark, smoke() and the tests all run on these.
Pure numpy generators -- no model arithmetic lives here (the CPU restatement of the model is oracle/, test-only).
"""
from __future__ import annotations

import numpy as np

# ----------------------------------------------------------------------------
# synthetic parameters / inputs -- no trained weights exist
# ----------------------------------------------------------------------------


def _orthogonal(rs, shape, gain):
    a = rs.normal(0.0, 1.0, shape)
    u, _, v = np.linalg.svd(a, full_matrices=False)
    q = u if u.shape == shape else v
    return (gain * q).astype(np.float32)


def param_shapes(arch):
    """Theano parameter names -> shapes for the inference graph (App. B.5)."""
    S = {}
    S["enc_conv1.W"], S["enc_conv1.b"] = (128, 3, 5, 5), (128,)
    for i, (ci, co) in enumerate(((128, 256), (256, 512), (512, 1024))):
        S["enc_conv%d.W" % (i + 2)] = (co, ci, 5, 5)
        for s in ("beta", "gamma", "mean", "inv_std"):
            S["bnorm%d.%s" % (i + 2, s)] = (co,)
    S["enc_fc1.W"] = (16384, 1000)
    S["enc_mu.W"], S["enc_logsigma.W"] = (1000, 100), (1000, 100)
    for bn, n in (("bnorm_enc_fc1", 1000), ("mu_bnorm", 100), ("ls_bnorm", 100)):
        for s in ("beta", "gamma", "mean", "inv_std"):
            S[bn + "." + s] = (n,)
    if arch == "IAN_simple":
        S["l_dec_fc2.W"] = (100, 16384)
        for s in ("beta", "gamma", "mean", "inv_std"):
            S["bnorm_dec_fc2." + s] = (16384,)
        for i, (ci, co) in enumerate(((1024, 512), (512, 256), (256, 128))):
            S["dec_conv%d.W" % (i + 1)] = (ci, co, 5, 5)
            for s in ("beta", "gamma", "mean", "inv_std"):
                S["bnorm_dc%d.%s" % (i + 1, s)] = (co,)
        S["dec_out.W"] = (128, 3, 5, 5)
        return S
    S["l_dec_fc2.W"], S["l_dec_fc2.b"] = (100, 8192), (8192,)
    for m in ("l_IAF_mu", "l_IAF_ls"):
        for l in ("_input", "_output_W", "_output_D"):
            S[m + l + ".W"], S[m + l + ".b"] = (100, 100), (100,)
    chans = ((512, 512, "dec_conv2a", [0, 2]), (512, 256, "dec_conv3a", [0, 2, 3]), (256, 128, "dec_conv4a", [0, 2, 3]))
    for i, (ci, co, blk, scales) in enumerate(chans):
        S["dec_conv%d.W" % (i + 1)] = (ci, co, 5, 5)
        for j in range(3):
            for s in ("beta", "gamma", "mean", "inv_std"):
                S["%sbnorm%d.%s" % (blk, j, s)] = (co,)
        for nm in (blk, blk + "2"):
            S[nm + "W"] = (co, co, 3, 3)
            S[nm + "_coeff_base"] = (co,)
            for sc in scales:
                S[nm + ("_coeff_1x1" if sc == 0 else "_coeff_%d" % sc)] = (co,)
    S["dec_conv4.W"] = (128, 128, 5, 5)
    for s in ("beta", "gamma", "mean", "inv_std"):
        S["bnorm_dc4." + s] = (128,)
    for nm, ci in (("R", 128), ("G_a", 128), ("G_b", 2), ("B_a", 128), ("B_b", 4)):
        S[nm + "W"] = (2, ci, 3, 3)
        S[nm + "_coeff_base"] = (2,)
        for sc in (2, 3, 4):
            S[nm + "_coeff_%d" % sc] = (2,)
    return S


def make_params(arch, seed=1):
    """Seeded synthetic parameters, deliberately NON-trivial:
    W ~ N(0,0.02) (initmethod(0.02), IAN_simple.py:79); biases and BN beta
    ~ N(0,0.1); gamma ~ U(0.5,1.5); running mean ~ N(0,0.1); inv_std ~ U(0.5,2)
    so that folding / indexing bugs show; MDC coefficients 1/(1+len(scales))
    (layers.py:214) perturbed +-20% per filter; MADE W orthogonal (layers.py:771).
    A few layers are re-scaled so that saturating outputs (tanh/sigmoid) stay in
    their sensitive range with random weights.
    """
    rs = np.random.RandomState(seed)
    P = {}
    for name, shp in param_shapes(arch).items():
        if name.endswith(".gamma"):
            v = rs.uniform(0.5, 1.5, shp)
        elif name.endswith(".inv_std"):
            v = rs.uniform(0.5, 2.0, shp)
        elif name.endswith(".mean") or name.endswith(".beta") or name.endswith(".b"):
            v = rs.normal(0, 0.1, shp)
        elif "_coeff_" in name:
            nsc = 3 if name.split("_coeff_")[0] in ("R", "G_a", "G_b", "B_a", "B_b", "dec_conv3a", "dec_conv3a2",
                                                     "dec_conv4a", "dec_conv4a2") else 2
            v = (1.0 / (1 + nsc)) * rs.uniform(0.8, 1.2, shp)
        elif name.startswith("l_IAF") and name.endswith(".W"):
            v = _orthogonal(rs, shp, np.sqrt(2.0)) * 0.3
        else:
            v = rs.normal(0, 0.02, shp)
        P[name] = np.asarray(v, np.float32)
    # keep deep random nets in a sane numeric range
    if arch == "IAN_simple":
        P["dec_out.W"] *= 1.0
    else:
        for nm in ("R", "G_a", "B_a"):
            P[nm + "W"] *= 4.0
        for nm in ("G_b", "B_b"):
            P[nm + "W"] *= 25.0
        for blk in ("dec_conv2a", "dec_conv3a", "dec_conv4a"):
            for nm in (blk, blk + "2"):
                P[nm + "W"] *= 0.5
        P["dec_conv4.W"] *= 0.5
    return P


def make_images(n, seed=0):
    """uint8 RandomState(seed).randint(0,256) low-pass filtered, then
    to_tanh (NPE.py:37-38). Returns float32 (n,3,64,64) in [-1,1]."""
    from scipy.ndimage import gaussian_filter
    rs = np.random.RandomState(seed)
    raw = rs.randint(0, 256, (n, 3, 64, 64)).astype(np.float32)
    sm = gaussian_filter(raw, sigma=(0, 0, 2, 2))
    sm = (sm - sm.min()) / (sm.max() - sm.min()) * 255.0
    img = np.uint8(sm)
    return (2.0 * (img.astype(np.float32) / 255.0) - 1.0).astype(np.float32)


def make_latents(n, seed=2):
    """NPE.py:319: np.random.randn(n,100) float32."""
    return np.random.RandomState(seed).randn(n, 100).astype(np.float32)


def train_param_shapes():
    """Shapes of the training-only parameters (discriminator head), IAN.py:209-216, layers.py:487-495."""
    return {"minibatch_discrim.theta": (1024, 500, 5), "minibatch_discrim.log_weight_scale": (500, 5),
            "minibatch_discrim.b": (500,), "discrimi.W": (1524, 3)}


def make_train_params(P, seed=3):
    """Adds seeded discriminator-head parameters to an inference parameter dict (oracle.make_params('IAN'))."""
    rs = np.random.RandomState(seed)
    Q = dict(P)
    Q["minibatch_discrim.theta"] = rs.normal(0, 0.05, (1024, 500, 5)).astype(np.float32)      # layers.py:487
    Q["minibatch_discrim.log_weight_scale"] = rs.normal(0, 0.1, (500, 5)).astype(np.float32)  # Constant(0) perturbed
    Q["minibatch_discrim.b"] = (-1.0 + rs.normal(0, 0.1, (500,))).astype(np.float32)          # Constant(-1) perturbed
    Q["discrimi.W"] = rs.normal(0, 0.02, (1524, 3)).astype(np.float32)
    return Q
