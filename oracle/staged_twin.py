"""CPU ORACLE (training step, staged) -- test infrastructure only, never imported by the product path.

``StagedTwin`` is ``train_twin.TrainTwin`` (the torch restatement of train_IAN.py:47-352, pinned against the
reference-executed fixture by tests/test_reference_pinned.py) with every intermediate tensor of the five network passes
NAMED, so that a run can either

* RECORD them (``provider=None``): ``self.rec[(pass, name)]`` -- used to check that staging changes nothing, and to play
  the role of a float32 implementation in the CPU self-test; or
* be evaluated AT another implementation's forward point (``provider(pass, name) -> tensor``): each stage is computed in this
  twin's precision from the PROVIDED inputs, its output is compared with the provided output (``self.local_err``: the LOCAL
  forward error of that stage, free of everything upstream) and then replaced by it with a straight-through estimator
  (value = provided, derivative = this twin's).  The piecewise-linear units (leaky ReLU, ReLU) take their branch from the
  PROVIDED post-activation value.  Autograd of the result is the exact (float64) gradient of the training loss at the other
  implementation's own activations and its own kink decisions.

Why (round-3 verdict, weak #1 / next-round item 1b): the composed gradient comparison "HIP float32 step vs float64 twin" mixes
two things -- (a) how far the float32 FORWARD drifts from the float64 forward, amplified by the conditioning of the graph
(batch statistics over 4 near-identical decoder outputs, |a_b - a_b'| kernels, leaky-ReLU kinks: scripts/exp/
fp32_noise_conditioning.py shows 1-ulp perturbations of the layer outputs moving single gradient tensors by 1e-2..6e-2),
and (b) the arithmetic error of the BACKWARD kernels.  The decomposition isolates (b): it is what a backward bug would show
up in, and it can be held to a tight bar.

Passes: EX / EH / EG = encoder(X / X_hat / X_gen), ZS = latent path, DZ / DG = decoder(z / IAF(Z)).
Names: encoder a1 y2 a2 y3 a3 y4 a4 feat act mbf p; latent y_fc1 f y_mu mu y_ls ls z0 z; decoder h0, <blk>_x _a _b _c _e _h for
blk in dec_conv2a / 3a / 4a, y4 h4 R G B xhat.  Layout: NCHW / (n, features), real channels only.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

from .train_twin import BN_EPS, DEC_STAGES, ENC_PARAMS, Z_PARAMS, TrainTwin, ortho_res


def _rel(a, b):
    a, b = a.detach().double(), b.detach().double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


class StagedTwin(TrainTwin):
    def __init__(self, *a, **kw):
        TrainTwin.__init__(self, *a, **kw)
        self.provider = None
        self.rec, self.local_err = {}, {}

    # ---- staging primitives ---------------------------------------------------------------------------------------
    def _get(self, tag, name):
        v = self.provider(tag, name)
        return v if torch.is_tensor(v) else torch.as_tensor(np.asarray(v))

    def stage(self, tag, name, v):
        """record v, or replace it (straight-through) by the provided value after noting the local error"""
        if self.provider is None:
            self.rec[(tag, name)] = v.detach().clone()
            return v
        h = self._get(tag, name).to(self.dtype)
        assert h.shape == v.shape, (tag, name, tuple(h.shape), tuple(v.shape))
        self.local_err[(tag, name)] = _rel(v, h)      # provided output vs this twin's stage applied to the provided input
        return v + (h - v).detach()

    def act(self, tag, name, pre, slope):
        """leaky ReLU (slope 0.2) / ReLU (slope 0) whose branch is the PROVIDED output's (a > 0 <=> pre-activation > 0)"""
        if self.provider is None:
            return self.stage(tag, name, torch.where(pre > 0, pre, slope * pre))
        h = self._get(tag, name).to(self.dtype)
        return self.stage(tag, name, torch.where(h > 0, pre, slope * pre))

    def bn_named(self, x, name):
        axes = [0] + list(range(2, x.ndim))
        mean = x.mean(axes, keepdim=True)
        var = ((x - mean) ** 2).mean(axes, keepdim=True)
        shp = (1, -1) + (1,) * (x.ndim - 2)
        return (x - mean) / torch.sqrt(var + BN_EPS) * self.P[name + ".gamma"].reshape(shp) + self.P[name + ".beta"].reshape(shp)

    # ---- the passes (train_twin.TrainTwin, stage by stage) ------------------------------------------------------------
    def encoder_s(self, tag, x):
        P = self.P
        h = self.act(tag, "a1", F.conv2d(x, P["enc_conv1.W"], P["enc_conv1.b"], stride=2, padding=2), 0.2)
        out = [h]
        for i in (2, 3, 4):
            y = self.stage(tag, "y%d" % i, F.conv2d(h, P["enc_conv%d.W" % i], None, stride=2, padding=2))
            h = self.act(tag, "a%d" % i, self.bn_named(y, "bnorm%d" % i), 0.2)
            out.append(h)
        return out

    def discriminator_s(self, tag, h4):
        P = self.P
        feat = self.stage(tag, "feat", h4.mean((2, 3)))                                   # GlobalPoolLayer
        theta, lws, b = P["minibatch_discrim.theta"], P["minibatch_discrim.log_weight_scale"], P["minibatch_discrim.b"]
        W = theta * (torch.exp(lws) / torch.sqrt((theta ** 2).sum(0))).unsqueeze(0)
        act = self.stage(tag, "act", torch.tensordot(feat, W, dims=([1], [0])))           # (B,500,5)
        n = feat.shape[0]
        abs_dif = (act.unsqueeze(3) - act.permute(1, 2, 0).unsqueeze(0)).abs().sum(2) + 1e6 * torch.eye(n, dtype=feat.dtype).unsqueeze(1)
        f = self.stage(tag, "mbf", torch.exp(-abs_dif).sum(2) + b.unsqueeze(0))           # layers.py:507-520
        return self.stage(tag, "p", torch.softmax(torch.cat([feat, f], 1) @ P["discrimi.W"], 1))

    def latent_s(self, h4, eps):
        P, t = self.P, "ZS"
        y = self.stage(t, "y_fc1", h4.flatten(1) @ P["enc_fc1.W"])
        f = self.act(t, "f", self.bn_named(y, "bnorm_enc_fc1"), 0.0)
        mu = self.stage(t, "mu", self.bn_named(self.stage(t, "y_mu", f @ P["enc_mu.W"]), "mu_bnorm"))
        ls = self.stage(t, "ls", self.bn_named(self.stage(t, "y_ls", f @ P["enc_logsigma.W"]), "ls_bnorm"))
        z0 = self.stage(t, "z0", mu + torch.exp(ls) * eps)
        return mu, ls, z0, self.stage(t, "z", self.iaf(z0))

    def decoder_s(self, tag, z):
        P = self.P
        h = self.act(tag, "h0", z @ P["l_dec_fc2.W"] + P["l_dec_fc2.b"], 0.2).reshape(-1, 512, 4, 4)
        for dc, blk, sc in DEC_STAGES:
            x = self.stage(tag, blk + "_x", self.deconv(h, dc))
            a = self.act(tag, blk + "_a", self.bn_named(x, blk + "bnorm0"), 0.2)
            b = self.stage(tag, blk + "_b", self.mdcl(a, blk, sc))
            c = self.act(tag, blk + "_c", self.bn_named(b, blk + "bnorm1"), 0.2)
            e = self.stage(tag, blk + "_e", x + self.mdcl(c, blk + "2", sc))
            h = self.act(tag, blk + "_h", self.bn_named(e, blk + "bnorm2"), 0.2)
        y4 = self.stage(tag, "y4", self.deconv(h, "dec_conv4"))
        h4 = self.act(tag, "h4", self.bn_named(y4, "bnorm_dc4"), 0.2)
        sc = [2, 3, 4]
        R = self.stage(tag, "R", torch.sigmoid(self.mdcl(h4, "R", sc)))
        G = self.stage(tag, "G", torch.sigmoid(self.mdcl(h4, "G_a", sc) + self.mdcl(R, "G_b", sc)))
        B = self.stage(tag, "B", torch.sigmoid(self.mdcl(h4, "B_a", sc) + self.mdcl(torch.cat([R, G], 1), "B_b", sc)))
        beta = lambda t: 2 * (t[:, 0:1] / (t[:, 0:1] + t[:, 1:2] + 1e-8)) - 1
        return self.stage(tag, "xhat", torch.cat([beta(R), beta(G), beta(B)], 1))

    # ---- the graph of train_IAN.py:116-250, staged ----------------------------------------------------------------------
    def losses_staged(self, X, Z, eps, provider=None, stop_xhat=False):
        """Same dictionary as TrainTwin.losses (minus the argmax metric)."""
        self.provider = provider
        self.rec, self.local_err = {}, {}
        c = self.cfg
        t = lambda a: torch.as_tensor(np.asarray(a), dtype=self.dtype) if not torch.is_tensor(a) else a.to(self.dtype)
        X, Z, eps = t(X), t(Z), t(eps)
        gX = self.encoder_s("EX", X)
        pX = self.discriminator_s("EX", gX[3])
        mu, ls, z0, z = self.latent_s(gX[3], eps)
        Xhat = self.decoder_s("DZ", z)
        Xhat_in = Xhat.detach() if stop_xhat else Xhat
        gXh = self.encoder_s("EH", Xhat_in)
        pXh = self.discriminator_s("EH", gXh[3])
        Xgen = self.decoder_s("DG", self.iaf(Z))
        gXg = self.encoder_s("EG", Xgen)
        pXg = self.discriminator_s("EG", gXg[3])
        ce = lambda p, k: (-torch.log(p[:, k])).mean()
        L = {}
        L["pixel_loss"] = (2 * (Xhat - X + 1e-8).abs()).mean()
        L["kl_div"] = -0.5 * (1 + 2 * ls - mu ** 2 - torch.exp(2 * ls)).mean()
        L["discrim_g_loss"] = ce(pXh, 1) + ce(pXg, 2)
        L["discrim_d_loss"] = ce(pX, 0)
        L["adv_discrim"] = c["dg_weight"] * L["discrim_g_loss"] + c["dd_weight"] * L["discrim_d_loss"]
        L["feature_loss"] = torch.stack([((a - b) ** 2).mean() for a, b in zip(gX, gXh)]).mean()
        L["gen_recon_loss"] = ce(pXh, 0)
        L["gen_sample_loss"] = ce(pXg, 0)
        L["adv_gen"] = c["agr_weight"] * L["gen_recon_loss"] + c["ags_weight"] * L["gen_sample_loss"]
        L["pixel_acc"] = 1 - ((Xhat - X) ** 2).mean()
        P = self.P
        reg_z = [n for n in Z_PARAMS if not n.endswith(".beta")]
        L["l2_Z"] = c["reg"] * sum((P[n] ** 2).sum() for n in reg_z)
        L["l2_discrim"] = c["ortho"] * ortho_res([(n, P[n]) for n in ENC_PARAMS])
        L["l2_gen"] = c["ortho"] * ortho_res([(n, P[n]) for n in self.groups["dec"]])
        self.provider = None
        return L

    def gradients_staged(self, X, Z, eps, which, provider=None):
        """-> ({group: {name: grad}}, losses) for update 'gen' (decoder_params + Z_params) or 'discrim' (encoder_params + Z_params),
        train_IAN.py:253-273."""
        c = self.cfg
        L = self.losses_staged(X, Z, eps, provider)
        err = dict(self.local_err)
        z_loss = c["feature_weight"] * L["feature_loss"] + c["recon_weight"] * L["pixel_loss"] + L["adv_gen"] + L["kl_div"] + L["l2_Z"]
        zp = [self.P[n] for n in self.groups["Z"]]
        out = {}
        if which == "gen":
            gen_loss = L["adv_gen"] + c["recon_weight"] * L["pixel_loss"] + c["feature_weight"] * L["feature_loss"] + L["l2_gen"]
            dec = [self.P[n] for n in self.groups["dec"]]
            out["dec"] = dict(zip(self.groups["dec"], torch.autograd.grad(gen_loss, dec, retain_graph=True)))
            out["Z"] = dict(zip(self.groups["Z"], torch.autograd.grad(z_loss, zp)))
        else:
            out["Z"] = dict(zip(self.groups["Z"], torch.autograd.grad(z_loss, zp)))
            Ld = self.losses_staged(X, Z, eps, provider, stop_xhat=True)           # consider_constant=[X_hat] (:253)
            enc = [self.P[n] for n in self.groups["enc"]]
            out["enc"] = dict(zip(self.groups["enc"], torch.autograd.grad(Ld["adv_discrim"] + Ld["l2_discrim"], enc)))
        self.local_err = err
        return out, {k: float(v) for k, v in L.items()}
