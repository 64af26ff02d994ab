"""CPU ORACLE (training step) -- test infrastructure only, never imported by the product path.

A torch-CPU restatement of ``make_training_functions`` (train_IAN.py:47-352) for the full IAN config
(IAN.py:67-228): the three network passes, the losses, the three parameter groups and the three
``lasagne.updates.adam`` instances.  Gradients come from torch autograd, so this twin is independent of the
hand-written backward kernels it checks.  PINNED against the reference-executed train_IAN.make_training_functions (tests/golden/ref_train_IAN.npz,
tests/test_reference_pinned.py); third-party primitive conventions [recalled] (see ian_oracle.py): Theano/Lasagne cannot be run
here; Lasagne semantics are [recalled].

Restated pieces and their reference lines:
  * batch_norm in training mode: batch mean / biased variance over all axes but 1, eps 1e-4 (App. B.3);
    every ``get_output`` call (train_IAN.py:116,140,149) normalises with its OWN batch statistics;
  * GaussianSampleLayer, stochastic: mu + exp(logsigma) * eps (layers.py:433); eps is an input here;
  * MADE x2 + IAFLayer (layers.py:641-650, 735-853) -- never trained (not in any parameter group);
  * MinibatchLayer (layers.py:486-524) on GlobalPool(enc_conv4) + 3-way softmax ``discrimi`` (IAN.py:209-216);
  * losses train_IAN.py:158-250, parameter groups :184-194, updates :253-276, metrics :291-304.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

from .ian_oracle import made_masks

BN_EPS = 1e-4

CFG = dict(reg=1e-5, ortho=1e-3, recon_weight=3.0, feature_weight=1.0, dg_weight=1.0, dd_weight=1.0, agr_weight=1.0,
           ags_weight=1.0, beta1=0.5, learning_rate=0.0002)  # IAN.py:38-62

ENC_PARAMS = ["enc_conv1.W", "enc_conv1.b", "enc_conv2.W", "bnorm2.beta", "bnorm2.gamma", "enc_conv3.W", "bnorm3.beta",
              "bnorm3.gamma", "enc_conv4.W", "bnorm4.beta", "bnorm4.gamma", "minibatch_discrim.theta",
              "minibatch_discrim.log_weight_scale", "minibatch_discrim.b", "discrimi.W"]
Z_PARAMS = ["enc_fc1.W", "bnorm_enc_fc1.beta", "bnorm_enc_fc1.gamma", "enc_mu.W", "mu_bnorm.beta", "mu_bnorm.gamma",
            "enc_logsigma.W", "ls_bnorm.beta", "ls_bnorm.gamma"]
DEC_STAGES = (("dec_conv1", "dec_conv2a", [0, 2]), ("dec_conv2", "dec_conv3a", [0, 2, 3]), ("dec_conv3", "dec_conv4a", [0, 2, 3]))
HEAD = ("R", "G_a", "G_b", "B_a", "B_b")


def mdcl_param_names(name, scales):
    out = [name + "W", name + "_coeff_base"]
    for s in scales:
        out.append(name + ("_coeff_1x1" if s == 0 else "_coeff_%d" % s))
    return out


def decoder_param_names():
    """train_IAN.py:193: trainable params of l_out that are not under l_Z, in Lasagne's topological order."""
    names = ["l_dec_fc2.W", "l_dec_fc2.b"]
    for dc, blk, scales in DEC_STAGES:
        names.append(dc + ".W")
        names += [blk + "bnorm0.beta", blk + "bnorm0.gamma"] + mdcl_param_names(blk, scales)
        names += [blk + "bnorm1.beta", blk + "bnorm1.gamma"] + mdcl_param_names(blk + "2", scales)
        names += [blk + "bnorm2.beta", blk + "bnorm2.gamma"]
    names += ["dec_conv4.W", "bnorm_dc4.beta", "bnorm_dc4.gamma"]
    for h in HEAD:
        names += mdcl_param_names(h, [2, 3, 4])
    return names


from neural_photo_editor_amd.synthetic import make_train_params, train_param_shapes  # noqa: E402,F401


def ortho_res(params):
    """train_IAN.py:158-165."""
    s = 0
    for name, x in params:
        if name[-1] == "W" and x.ndim == 4:
            y = torch.einsum("abik,abjk->aij", x, x)
            y = y - torch.eye(x.shape[2], x.shape[3], dtype=x.dtype).unsqueeze(0)
            s = s + y.abs().sum()
    return s


class TrainTwin:
    def __init__(self, P, cfg=None, dtype=torch.float64, deconv_flip=True):
        self.cfg = dict(CFG)
        if cfg:
            self.cfg.update({k: v for k, v in cfg.items() if k in self.cfg and not isinstance(v, dict)})
        self.dtype = dtype
        self.flip = deconv_flip
        self.P = {k: torch.tensor(np.asarray(v), dtype=dtype) for k, v in P.items()}
        self.masks = tuple(torch.tensor(m, dtype=dtype) for m in made_masks())
        self.groups = {"enc": list(ENC_PARAMS), "Z": list(Z_PARAMS), "dec": decoder_param_names()}
        for g in self.groups.values():
            for n in g:
                assert n in self.P, n
                self.P[n].requires_grad_(True)
        self.adam = {g: {"t": 0, "m": {n: torch.zeros_like(self.P[n]) for n in names},
                         "v": {n: torch.zeros_like(self.P[n]) for n in names}} for g, names in self.groups.items()}
        self.lr = float(self.cfg["learning_rate"])

    # ---- layers (training mode) -------------------------------------------------------------------------
    def bn(self, x, name):
        axes = [0] + list(range(2, x.ndim))
        mean = x.mean(axes, keepdim=True)
        var = ((x - mean) ** 2).mean(axes, keepdim=True)
        shp = (1, -1) + (1,) * (x.ndim - 2)
        return (x - mean) / torch.sqrt(var + BN_EPS) * self.P[name + ".gamma"].reshape(shp) + self.P[name + ".beta"].reshape(shp)

    def deconv(self, x, name):
        W = self.P[name + ".W"]
        Wt = torch.flip(W, (2, 3)) if self.flip else W
        return F.conv_transpose2d(x, Wt, None, stride=2, padding=2, output_padding=1)

    def mdcl(self, x, name, scales):
        P = self.P
        W = P[name + "W"]
        out = F.conv2d(x, W, padding=1) * P[name + "_coeff_base"].reshape(1, -1, 1, 1)
        for s in scales:
            if s == 0:
                out = out + F.conv2d(x, W.mean((2, 3), keepdim=True)) * P[name + "_coeff_1x1"].reshape(1, -1, 1, 1)
            else:
                out = out + F.conv2d(x, W, padding=s, dilation=s) * P[name + "_coeff_%d" % s].reshape(1, -1, 1, 1)
        return out

    def mdblock(self, x, name, scales):
        lr = lambda t: F.leaky_relu(t, 0.2)
        a = lr(self.bn(x, name + "bnorm0"))
        c = lr(self.bn(self.mdcl(a, name, scales), name + "bnorm1"))
        d = self.mdcl(c, name + "2", scales)
        return lr(self.bn(x + d, name + "bnorm2"))

    def made(self, z, name):
        P, (M0, M1, MD) = self.P, self.masks
        h = torch.relu(z @ (P[name + "_input.W"] * M0) + P[name + "_input.b"])
        return (h @ (P[name + "_output_W.W"] * M1) + P[name + "_output_W.b"]) + (z @ (P[name + "_output_D.W"] * MD) + P[name + "_output_D.b"])

    def iaf(self, z0):
        # the reference graph feeds each MADE with its own first masked layer's output (ian_oracle.made_as_wired)
        hid = lambda n: torch.relu(z0 @ (self.P[n + "_input.W"] * self.masks[0]) + self.P[n + "_input.b"])
        return (z0 - self.made(hid("l_IAF_mu"), "l_IAF_mu")) / torch.exp(self.made(hid("l_IAF_ls"), "l_IAF_ls"))

    def encoder(self, x):
        """-> (introspection features [4], enc_conv4 output)"""
        P = self.P
        lr = lambda t: F.leaky_relu(t, 0.2)
        h1 = lr(F.conv2d(x, P["enc_conv1.W"], P["enc_conv1.b"], stride=2, padding=2))
        h2 = lr(self.bn(F.conv2d(h1, P["enc_conv2.W"], None, stride=2, padding=2), "bnorm2"))
        h3 = lr(self.bn(F.conv2d(h2, P["enc_conv3.W"], None, stride=2, padding=2), "bnorm3"))
        h4 = lr(self.bn(F.conv2d(h3, P["enc_conv4.W"], None, stride=2, padding=2), "bnorm4"))
        return [h1, h2, h3, h4]

    def minibatch(self, feat):
        """layers.py:486-524 (init=False)."""
        P = self.P
        theta, lws, b = P["minibatch_discrim.theta"], P["minibatch_discrim.log_weight_scale"], P["minibatch_discrim.b"]
        W = theta * (torch.exp(lws) / torch.sqrt((theta ** 2).sum(0))).unsqueeze(0)
        act = torch.tensordot(feat, W, dims=([1], [0]))                      # (B,500,5)
        n = feat.shape[0]
        abs_dif = (act.unsqueeze(3) - act.permute(1, 2, 0).unsqueeze(0)).abs().sum(2) \
            + 1e6 * torch.eye(n, dtype=feat.dtype).unsqueeze(1)              # (B,500,B)
        f = torch.exp(-abs_dif).sum(2) + b.unsqueeze(0)
        return torch.cat([feat, f], 1)

    def discriminator(self, h4):
        feat = h4.mean((2, 3))                                               # GlobalPoolLayer
        return torch.softmax(self.minibatch(feat) @ self.P["discrimi.W"], 1)

    def latent(self, h4, eps):
        P = self.P
        f = torch.relu(self.bn(h4.flatten(1) @ P["enc_fc1.W"], "bnorm_enc_fc1"))
        mu = self.bn(f @ P["enc_mu.W"], "mu_bnorm")
        ls = self.bn(f @ P["enc_logsigma.W"], "ls_bnorm")
        return mu, ls, mu + torch.exp(ls) * eps

    def decoder(self, z):
        P = self.P
        h = F.leaky_relu(z @ P["l_dec_fc2.W"] + P["l_dec_fc2.b"], 0.2).reshape(-1, 512, 4, 4)
        for dc, blk, sc in DEC_STAGES:
            h = self.mdblock(self.deconv(h, dc), blk, sc)
        h = F.leaky_relu(self.bn(self.deconv(h, "dec_conv4"), "bnorm_dc4"), 0.2)
        sc = [2, 3, 4]
        R = torch.sigmoid(self.mdcl(h, "R", sc))
        G = torch.sigmoid(self.mdcl(h, "G_a", sc) + self.mdcl(R, "G_b", sc))
        B = torch.sigmoid(self.mdcl(h, "B_a", sc) + self.mdcl(torch.cat([R, G], 1), "B_b", sc))
        beta = lambda t: 2 * (t[:, 0:1] / (t[:, 0:1] + t[:, 1:2] + 1e-8)) - 1
        return torch.cat([beta(R), beta(G), beta(B)], 1)

    # ---- the graph of train_IAN.py:116-250 ---------------------------------------------------------------
    def losses(self, X, Z, eps, stop_xhat=False):
        c = self.cfg
        t = lambda a: torch.as_tensor(np.asarray(a), dtype=self.dtype) if not torch.is_tensor(a) else a
        X, Z, eps = t(X), t(Z), t(eps)
        gX = self.encoder(X)
        pX = self.discriminator(gX[3])
        mu, ls, z0 = self.latent(gX[3], eps)
        Xhat = self.decoder(self.iaf(z0))
        Xhat_in = Xhat.detach() if stop_xhat else Xhat                       # consider_constant=[X_hat] (:253)
        gXh = self.encoder(Xhat_in)
        pXh = self.discriminator(gXh[3])
        Xgen = self.decoder(self.iaf(Z))
        pXg = self.discriminator(self.encoder(Xgen)[3])
        ce = lambda p, k: (-torch.log(p[:, k])).mean()                       # categorical_crossentropy with one-hot targets
        L = {}
        L["pixel_loss"] = (2 * (Xhat - X + 1e-8).abs()).mean()               # :169
        L["kl_div"] = -0.5 * (1 + 2 * ls - mu ** 2 - torch.exp(2 * ls)).mean()   # :172
        L["discrim_g_loss"] = ce(pXh, 1) + ce(pXg, 2)                        # :228 (p2=[0,1,0], p3=[0,0,1]; :482-484)
        L["discrim_d_loss"] = ce(pX, 0)                                      # :234
        L["adv_discrim"] = c["dg_weight"] * L["discrim_g_loss"] + c["dd_weight"] * L["discrim_d_loss"]
        acc = lambda p, k: (p.argmax(1) == k).to(self.dtype).mean()
        L["discrim_acc"] = (acc(pX, 0) + acc(pXh, 1) + acc(pXg, 2)) / 3.0      # :240
        L["feature_loss"] = torch.stack([((a - b) ** 2).mean() for a, b in zip(gX, gXh)]).mean()   # :244
        L["gen_recon_loss"] = ce(pXh, 0)                                     # :247
        L["gen_sample_loss"] = ce(pXg, 0)                                    # :248
        L["adv_gen"] = c["agr_weight"] * L["gen_recon_loss"] + c["ags_weight"] * L["gen_sample_loss"]
        L["pixel_acc"] = 1 - ((Xhat - X) ** 2).mean()                        # :279
        P = self.P
        reg_z = [n for n in Z_PARAMS if not n.endswith(".beta")]             # regularizable: W and gamma
        L["l2_Z"] = c["reg"] * sum((P[n] ** 2).sum() for n in reg_z)          # :211-213
        L["l2_discrim"] = c["ortho"] * ortho_res([(n, P[n]) for n in ENC_PARAMS])           # :214-218
        L["l2_gen"] = c["ortho"] * ortho_res([(n, P[n]) for n in self.groups["dec"]])      # :219-221
        self.tensors = {"X_hat": Xhat, "X_gen": Xgen, "mu": mu, "ls": ls, "z0": z0, "p_X": pX, "p_X_hat": pXh, "p_X_gen": pXg,
                        "g_X": gX, "g_X_hat": gXh}
        return L

    def gradients(self, X, Z, eps):
        """-> dict group -> {name: grad} for the three update rules (train_IAN.py:253-273)."""
        c = self.cfg
        L = self.losses(X, Z, eps)
        gen_loss = L["adv_gen"] + c["recon_weight"] * L["pixel_loss"] + c["feature_weight"] * L["feature_loss"] + L["l2_gen"]
        z_loss = c["feature_weight"] * L["feature_loss"] + c["recon_weight"] * L["pixel_loss"] + L["adv_gen"] + L["kl_div"] + L["l2_Z"]
        dec = [self.P[n] for n in self.groups["dec"]]
        zp = [self.P[n] for n in self.groups["Z"]]
        g_dec = torch.autograd.grad(gen_loss, dec, retain_graph=True)
        g_z = torch.autograd.grad(z_loss, zp)
        Ld = self.losses(X, Z, eps, stop_xhat=True)
        enc = [self.P[n] for n in self.groups["enc"]]
        g_enc = torch.autograd.grad(Ld["adv_discrim"] + Ld["l2_discrim"], enc)
        metrics = {k: float(v) for k, v in L.items()}
        return {"dec": dict(zip(self.groups["dec"], g_dec)), "Z": dict(zip(self.groups["Z"], g_z)),
                "enc": dict(zip(self.groups["enc"], g_enc))}, metrics

    def _adam(self, group, grads):
        """lasagne.updates.adam (App. B.7), beta2 = 0.999, epsilon = 1e-8."""
        st = self.adam[group]
        b1, b2, e = self.cfg["beta1"], 0.999, 1e-8
        st["t"] += 1
        t = st["t"]
        a_t = self.lr * np.sqrt(1 - b2 ** t) / (1 - b1 ** t)
        with torch.no_grad():
            for n, g in grads.items():
                st["m"][n] = b1 * st["m"][n] + (1 - b1) * g
                st["v"][n] = b2 * st["v"][n] + (1 - b2) * g * g
                self.P[n] -= a_t * st["m"][n] / (torch.sqrt(st["v"][n]) + e)

    def update_gen(self, X, Z, eps):
        """train_IAN.py:309-318: returns [gen_recon_loss, gen_sample_loss, pixel_loss, feature_loss, pixel_acc]."""
        g, m = self.gradients(X, Z, eps)
        self._adam("dec", g["dec"])
        self._adam("Z", g["Z"])
        return [m[k] for k in ("gen_recon_loss", "gen_sample_loss", "pixel_loss", "feature_loss", "pixel_acc")]

    def update_discrim(self, X, Z, eps):
        """train_IAN.py:320-329: returns [discrim_g_loss, discrim_d_loss, discrim_acc, pixel_loss, pixel_acc]."""
        g, m = self.gradients(X, Z, eps)
        self._adam("enc", g["enc"])
        self._adam("Z", g["Z"])
        return [m[k] for k in ("discrim_g_loss", "discrim_d_loss", "discrim_acc", "pixel_loss", "pixel_acc")]

    def numpy_params(self):
        return {k: v.detach().numpy().astype(np.float32) for k, v in self.P.items()}
