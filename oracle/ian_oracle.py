"""CPU ORACLE -- test infrastructure only, never shipped, never measured as the product.

A float32/float64 numpy restatement of the Introspective Adversarial Network
inference graphs of ajbrock/Neural-Photo-Editor, written from the reference's
model definitions.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import this package.

PINNING: the reference-OWNED arithmetic is pinned by execution -- the reference's
layers.py / mask_generator.py / IAN.py / IAN_simple.py / API.py / train_IAN.py /
sample_IAN.py / GANcheckpoints.py run UNMODIFIED on the evaluating Theano/Lasagne
stand-in of ``oracle/refexec`` and write tests/golden/ref_*.npz
(tests/golden/make_ref_golden.py); tests/test_reference_pinned.py holds this
restatement to those fixtures at float64 round-off.  What stays "[recalled]" is
the third-party primitive layer (Theano "development version" + Lasagne,
README.md:11-12, cannot be imported here; no trained weights or images exist,
SURVEY.md section 0 / 8c): conv / transposed-conv / dilated-conv / batch_norm /
adam conventions and RandomStreams seeding, restated once in
oracle/refexec/minilasagne.py and once here.

Every function cites the reference lines it restates (paths relative to
/root/reference).  Lasagne/Theano primitive semantics are "[recalled]" from the
public sources.

Tensor convention here is the reference's: NCHW float arrays.
"""
from __future__ import annotations

import hashlib

import numpy as np

BN_EPS = 1e-4  # lasagne.layers.BatchNormLayer default epsilon [recalled]

# ----------------------------------------------------------------------------
# nonlinearities (lasagne.nonlinearities) [recalled], App. B.4
# ----------------------------------------------------------------------------


def act_fn(name, x):
    if name in (None, "identity", "linear"):
        return x
    if name == "relu":
        return np.maximum(x, 0)
    if name == "lrelu":  # LeakyRectify(0.2): IAN_simple.py:80, IAN.py:77
        return np.where(x > 0, x, x * x.dtype.type(0.2))
    if name == "elu":  # IAN_simple.py:121
        return np.where(x > 0, x, np.expm1(np.minimum(x, 0)))
    if name == "tanh":  # IAN_simple.py:179
        return np.tanh(x)
    if name == "sigmoid":  # IAN.py:186
        return 1.0 / (1.0 + np.exp(-x))
    raise ValueError(name)


# ----------------------------------------------------------------------------
# primitives
# ----------------------------------------------------------------------------


def conv5s2(x, W, b=None):
    """5x5 stride-2 pad-2 cross-correlation (flip_filters=False).

    IAN_simple.py:73-116 / IAN.py:71-110 (Conv2DDNNLayer / Conv2DLayer with
    flip_filters=False).  x (N,Cin,H,W), W (Cout,Cin,5,5) -> (N,Cout,H/2,W/2)
    out[n,co,oy,ox] = sum W[co,ci,ky,kx] * x[n,ci,2oy-2+ky,2ox-2+kx]
    """
    N, C, H, Wd = x.shape
    OH, OW = H // 2, Wd // 2
    xp = np.zeros((N, C, H + 4, Wd + 4), x.dtype)
    xp[:, :, 2:2 + H, 2:2 + Wd] = x
    out = np.zeros((N, W.shape[0], OH, OW), x.dtype)
    for ky in range(5):
        for kx in range(5):
            patch = xp[:, :, ky:ky + 2 * OH:2, kx:kx + 2 * OW:2]  # N,C,OH,OW
            out += np.einsum("nchw,oc->nohw", patch, W[:, :, ky, kx], optimize=True)
    if b is not None:
        out += b[None, :, None, None]
    return out


def deconv5s2(x, W, b=None, flip=True):
    """5x5 stride-2 transposed convolution with output forced to 2x input.

    layers.py:436-483 (DeconvLayer: GpuDnnConvGradI of a conv_mode='conv',
    border_mode=crop=2, subsample=2 descriptor; output shape = 2x, :460).
    W (Cin,Cout,5,5) (:449-452).  Scatter rule:
      out[n,co,oy,ox] += x[n,ci,iy,ix] * W[ci,co,kt_y,kt_x],  oy = 2iy-2+ky
    with kt = 4-k when ``flip`` (gradient of a true convolution, App. B.2).
    The same rule is the dnn=False branch of IAN_simple.py:182-223
    (TransposedConv2DLayer crop=1 then drop first row/col).
    """
    N, C, H, Wd = x.shape
    Cout = W.shape[1]
    full = np.zeros((N, Cout, 2 * H + 3, 2 * Wd + 3), x.dtype)
    for ky in range(5):
        for kx in range(5):
            wk = W[:, :, 4 - ky, 4 - kx] if flip else W[:, :, ky, kx]  # Cin,Cout
            full[:, :, ky:ky + 2 * H:2, kx:kx + 2 * Wd:2] += np.einsum(
                "nchw,co->nohw", x, wk, optimize=True)
    out = full[:, :, 2:2 + 2 * H, 2:2 + 2 * Wd]
    if b is not None:
        out = out + b[None, :, None, None]
    return np.ascontiguousarray(out)


def dilated_corr3(x, W, d):
    """3x3 correlation with dilation d on input zero-padded by d (same size).

    d=1 is layers.py:223-232 (base conv, pad 1); d>1 is layers.py:250-257
    (PadLayer(width=d) + DilatedConv2DLayer(dilation=d), which correlates,
    [recalled]).  W (Cout,Cin,3,3).
    """
    N, C, H, Wd = x.shape
    xp = np.zeros((N, C, H + 2 * d, Wd + 2 * d), x.dtype)
    xp[:, :, d:d + H, d:d + Wd] = x
    out = np.zeros((N, W.shape[0], H, Wd), x.dtype)
    for p in range(3):
        for q in range(3):
            patch = xp[:, :, p * d:p * d + H, q * d:q * d + Wd]
            out += np.einsum("nchw,oc->nohw", patch, W[:, :, p, q], optimize=True)
    return out


def mdcl(x, P, name, scales):
    """Multiscale Dilated Convolution layer, layers.py:207-258.

    Sum (ElemwiseSumLayer, :258) of branches that share one W (:220):
      base 3x3 pad 1 scaled per output filter by <name>_coeff_base (:223-232);
      scale 0: 1x1 conv with mean(W,[2,3]) * <name>_coeff_1x1 (:238-247);
      scale s>0: dilated 3x3 * <name>_coeff_<s> (:250-257).
    No bias, no nonlinearity.
    """
    W = P[name + "W"]
    out = dilated_corr3(x, W, 1) * P[name + "_coeff_base"][None, :, None, None]
    for s in scales:
        if s == 0:
            w1 = W.mean(axis=(2, 3))  # Cout,Cin
            y = np.einsum("nchw,oc->nohw", x, w1, optimize=True)
            out = out + y * P[name + "_coeff_1x1"][None, :, None, None]
        else:
            out = out + dilated_corr3(x, W, s) * P[name + "_coeff_%d" % s][None, :, None, None]
    return out


def bn_inf(x, P, name):
    """Inference BatchNormLayer: (x-mean)*(gamma*inv_std)+beta, App. B.3.

    Per channel for 4-D input, per feature for 2-D input (axes = all but 1).
    """
    shp = (1, -1) + (1,) * (x.ndim - 2)
    g, bta = P[name + ".gamma"].reshape(shp), P[name + ".beta"].reshape(shp)
    m, s = P[name + ".mean"].reshape(shp), P[name + ".inv_std"].reshape(shp)
    return (x - m) * (g * s) + bta


def dense(x, W, b=None):
    """DenseLayer: flatten trailing dims row-major (C,H,W) then x.W + b, App. B.6."""
    x2 = x.reshape(x.shape[0], -1)
    y = x2 @ W
    if b is not None:
        y = y + b[None, :]
    return y


def made(z, P, name, masks):
    """MADE with hidden_sizes=[100] and direct input->output connection.

    layers.py:735-853: m = relu(z.(W0*M0)+b0) (:775-781, MaskedLayer :653-674);
    out = (m.(W1*M1)+b1) + (z.(WD*MD)+bD) (:797-812, DIML :680-707).
    """
    M0, M1, MD = masks
    h = act_fn("relu", z @ (P[name + "_input.W"] * M0) + P[name + "_input.b"])
    o = h @ (P[name + "_output_W.W"] * M1) + P[name + "_output_W.b"]
    d = z @ (P[name + "_output_D.W"] * MD) + P[name + "_output_D.b"]
    return o + d


def made_as_wired(z, P, name, masks):
    """What ``lasagne.layers.get_output`` actually feeds through a reference MADE layer.

    layers.py:775 assigns ``self.input_layer = MaskedLayer(incoming=z, ...)``, overwriting the attribute
    ``lasagne.layers.Layer.__init__`` set to ``z``.  Lasagne's ``get_all_layers`` / ``get_output`` follow
    ``layer.input_layer`` (helper.py: ``layer_inputs = all_outputs[layer.input_layer]``), so the MADE's
    ``get_output_for`` (layers.py:814-815) receives the OUTPUT of its own first masked layer,
    h = relu(z.(W0*M0)+b0), and evaluates the whole masked MLP on h:  made(h), not made(z).
    Established by executing layers.py on the evaluating stand-in (tests/golden/ref_layers.npz 'iaf/*',
    ref_IAN.npz 'z'); it is the reference's behaviour in API.py:50, sample_IAN.py:94 and train_IAN.py:116,149.
    """
    M0 = masks[0]
    h = act_fn("relu", z @ (P[name + "_input.W"] * M0) + P[name + "_input.b"])
    return made(h, P, name, masks)


def beta_layer(a, b):
    """layers.py:397-408: 2*(alpha/(alpha+beta+1e-8))-1."""
    return 2 * (a / (a + b + a.dtype.type(1e-8))) - 1


# ----------------------------------------------------------------------------
# MADE masks (bit-exact requirement), mask_generator.py:15-103, SURVEY App. C
# ----------------------------------------------------------------------------


def made_ordering(input_size=100, random_seed=1234):
    """Ordering after MADE.reset("Once") (layers.py:845-853, API.py:33-36).

    mask_generator.py:20,35: RandomStreams(seed).shuffle_row_elements(ordering).
    [recalled] theano.tensor.shared_randomstreams.RandomStreams seeds each
    random variable's RandomState with RandomState(seed).randint(2**30);
    reset() re-seeds (:60) so "Once" always yields the first permutation.
    shuffle_row_elements on a vector = RandomState.permutation applied to it.
    """
    child = int(np.random.RandomState(random_seed).randint(2 ** 30))
    perm = np.random.RandomState(child).permutation(input_size)
    # permute_row_elements: out[i] = ordering[perm[i]], ordering = arange
    return perm.astype(np.int64), child


def made_masks(input_size=100, hidden=100, random_seed=1234):
    """The three 0/1 masks for MADE(hidden_sizes=[100]), mask_distribution=0.

    mask_generator.py:75-91: with l=0 the multinomial is one-hot, every hidden
    unit gets connectivity min(ordering+1) = 1.
    _get_mask (:93-94): M[i,j] = (c_in[i] <= c_out[j]).
      M0 = mask(layer 0 -> 1): c_in = ordering+1, c_out = hidden connectivity
      M1 = mask(layer 1 -> 2): c_in = hidden connectivity, c_out = ordering
      MD = direct mask(0 -> 2): c_in = ordering+1, c_out = ordering   (:99-100)
    Returned as float32 arrays of exact 0.0/1.0 (layers.py:660,690).
    """
    ordering, _ = made_ordering(input_size, random_seed)
    c_in0 = ordering + 1
    c_hid = np.full((hidden,), int(c_in0.min()), np.int64)
    c_out = ordering
    M0 = (c_in0[:, None] <= c_hid[None, :]).astype(np.float32)
    M1 = (c_hid[:, None] <= c_out[None, :]).astype(np.float32)
    MD = (c_in0[:, None] <= c_out[None, :]).astype(np.float32)
    return M0, M1, MD


def ordering_digest(ordering):
    return hashlib.sha256(np.asarray(ordering, np.int64).tobytes()).hexdigest()[:16]


# ----------------------------------------------------------------------------
# model graphs
# ----------------------------------------------------------------------------

ARCHS = ("IAN_simple", "IAN")


class Oracle:
    """Restated inference graphs.  ``P`` maps Theano parameter names (App. B.5)
    to arrays; dtype of ``P`` decides float32 vs the float64 twin."""

    def __init__(self, arch, P, deconv_flip=True, dtype=np.float32):
        assert arch in ARCHS
        self.arch = arch
        self.dtype = np.dtype(dtype)
        self.P = {k: np.asarray(v, self.dtype) for k, v in P.items()}
        self.flip = deconv_flip
        self.masks = tuple(m.astype(self.dtype) for m in made_masks()) if arch == "IAN" else None

    # -- encoder: IAN_simple.py:73-126 / IAN.py:71-126 ------------------------
    def encoder_features(self, x):
        P = self.P
        x = np.asarray(x, self.dtype)
        h1 = act_fn("lrelu", conv5s2(x, P["enc_conv1.W"], P["enc_conv1.b"]))
        h2 = act_fn("lrelu", bn_inf(conv5s2(h1, P["enc_conv2.W"]), P, "bnorm2"))
        h3 = act_fn("lrelu", bn_inf(conv5s2(h2, P["enc_conv3.W"]), P, "bnorm3"))
        h4 = act_fn("lrelu", bn_inf(conv5s2(h3, P["enc_conv4.W"]), P, "bnorm4"))
        return [h1, h2, h3, h4]

    def Zfn(self, x):
        """x -> deterministic latent before IAF (mu).  sample_IAN.py:90-91;
        GaussianSampleLayer deterministic returns mu (layers.py:431-432)."""
        P = self.P
        h4 = self.encoder_features(x)[-1]
        fc_act = "elu" if self.arch == "IAN_simple" else "relu"  # IAN_simple.py:121 / IAN.py:118
        f = act_fn(fc_act, bn_inf(dense(h4, P["enc_fc1.W"]), P, "bnorm_enc_fc1"))
        mu = bn_inf(dense(f, P["enc_mu.W"]), P, "mu_bnorm")
        return mu

    def Z_IAF_fn(self, z):
        """IAN.py:127-128, layers.py:641-650: (z - mu_IAF)/exp(ls_IAF)."""
        if self.arch == "IAN_simple":
            return np.asarray(z, self.dtype)
        z = np.asarray(z, self.dtype)
        m = made_as_wired(z, self.P, "l_IAF_mu", self.masks)   # see made_as_wired: MADE evaluated on its own
        s = made_as_wired(z, self.P, "l_IAF_ls", self.masks)   # first hidden layer, as the reference graph does
        return (z - m) / np.exp(s)

    def encode_images(self, x):
        """API.py:78-90 -> Z_hat_fn (API.py:50-51)."""
        return self.Z_IAF_fn(self.Zfn(x))

    # -- decoder ---------------------------------------------------------------
    def decoder_activations(self, z):
        """Returns list of (name, activation) through the decoder; last is x_hat."""
        P = self.P
        z = np.asarray(z, self.dtype)
        acts = []
        if self.arch == "IAN_simple":
            # IAN_simple.py:129-139: dense -> BN(per feature) -> relu -> (1024,4,4)
            h = act_fn("relu", bn_inf(dense(z, P["l_dec_fc2.W"]), P, "bnorm_dec_fc2"))
            h = h.reshape(-1, 1024, 4, 4)
            acts.append(("dec_fc2", h))
            for i, nm in enumerate(("dec_conv1", "dec_conv2", "dec_conv3")):  # :141-170
                h = act_fn("relu", bn_inf(deconv5s2(h, P[nm + ".W"], None, self.flip), P, "bnorm_dc%d" % (i + 1)))
                acts.append((nm, h))
            h = act_fn("tanh", deconv5s2(h, P["dec_out.W"], None, self.flip))  # :171-181
            acts.append(("dec_out", h))
            return acts
        # full IAN, IAN.py:129-207
        h = act_fn("lrelu", dense(z, P["l_dec_fc2.W"], P["l_dec_fc2.b"])).reshape(-1, 512, 4, 4)
        acts.append(("dec_fc2", h))
        blocks = (("dec_conv1", "dec_conv2a", [0, 2]), ("dec_conv2", "dec_conv3a", [0, 2, 3]),
                  ("dec_conv3", "dec_conv4a", [0, 2, 3]))
        for dc, blk, scales in blocks:
            # DeconvLayer nonlinearity=None; its bias is removed by the
            # batch_norm() call inside MDBLOCK (layers.py:412, App. B.3: batch_norm
            # strips the wrapped layer's bias) -- parameters named <dc>.b do not exist.
            h = deconv5s2(h, P[dc + ".W"], P.get(dc + ".b"), self.flip)
            acts.append((dc, h))
            h = self.mdblock(h, blk, scales)
            acts.append((blk, h))
        h = act_fn("lrelu", bn_inf(deconv5s2(h, P["dec_conv4.W"], None, self.flip), P, "bnorm_dc4"))
        acts.append(("dec_conv4", h))
        sc = [2, 3, 4]
        R = act_fn("sigmoid", mdcl(h, P, "R", sc))  # IAN.py:183-186
        G = act_fn("sigmoid", mdcl(h, P, "G_a", sc) + mdcl(R, P, "G_b", sc))  # :187-196
        B = act_fn("sigmoid", mdcl(h, P, "B_a", sc) + mdcl(np.concatenate([R, G], 1), P, "B_b", sc))  # :197-206
        acts += [("R", R), ("G", G), ("B", B)]
        out = np.concatenate([beta_layer(R[:, 0:1], R[:, 1:2]), beta_layer(G[:, 0:1], G[:, 1:2]),
                              beta_layer(B[:, 0:1], B[:, 1:2])], 1)  # :207
        acts.append(("out", out))
        return acts

    def mdblock(self, x, name, scales):
        """layers.py:411-416 pre-activation residual block, nonlinearity lrelu(0.2)."""
        P = self.P
        a = act_fn("lrelu", bn_inf(x, P, name + "bnorm0"))
        b = mdcl(a, P, name, scales)
        c = act_fn("lrelu", bn_inf(b, P, name + "bnorm1"))
        d = mdcl(c, P, name + "2", scales)
        return act_fn("lrelu", bn_inf(x + d, P, name + "bnorm2"))

    def sample_at(self, z):
        """API.py:98-110 -> X_hat_fn (API.py:46-47): l_Z -> l_out, deterministic."""
        return self.decoder_activations(z)[-1][1]

    def sample(self, z_iaf):
        """sample_IAN.py:86: l_Z_IAF -> l_out."""
        return self.sample_at(self.Z_IAF_fn(z_iaf))

    def reconstruct(self, x):
        return self.sample_at(self.encode_images(x))


# ----------------------------------------------------------------------------
# synthetic parameters / inputs -- no trained weights exist.  The generators are plain numpy and live
# in the product package (bench.py uses them as its data source); re-exported here for the tests.
# ----------------------------------------------------------------------------
from neural_photo_editor_amd.synthetic import (make_images, make_latents, make_params, make_train_params,  # noqa: E402,F401
                                               param_shapes, train_param_shapes)
