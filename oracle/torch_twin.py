"""CPU ORACLE (torch-CPU twin) -- test infrastructure only.

The same restated graphs as ``ian_oracle.py`` but built from torch CPU ops so
that (i) the numpy restatement's index conventions can be cross-checked against
an independently implemented conv / conv_transpose, (ii) d(loss)/dz of
API.py:59,64 is available through autograd in float32 and float64, and (iii)
``bench.py``'s ``cpu_baseline`` leg has a multi-threaded CPU implementation to
time ("CPU restatement, not Theano").

PINNED against the reference-executed API.py gradients (tests/golden/ref_IAN*.npz); see ian_oracle.py header.  Never imported by the product path.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

from .ian_oracle import made_masks


def _act(name, x):
    if name in (None, "identity"):
        return x
    if name == "relu":
        return torch.relu(x)
    if name == "lrelu":
        return F.leaky_relu(x, 0.2)
    if name == "elu":
        return F.elu(x)
    if name == "tanh":
        return torch.tanh(x)
    if name == "sigmoid":
        return torch.sigmoid(x)
    raise ValueError(name)


class TorchTwin:
    def __init__(self, arch, P, deconv_flip=True, dtype=torch.float32):
        self.arch, self.dtype, self.flip = arch, dtype, deconv_flip
        self.P = {k: torch.as_tensor(np.asarray(v), dtype=dtype) for k, v in P.items()}
        self.masks = tuple(torch.as_tensor(m, dtype=dtype) for m in made_masks()) if arch == "IAN" else None
        # transposed-conv weights in torch orientation (App. B.2): flip(W,(2,3))
        self.Wt = {}
        for k, v in self.P.items():
            if k.startswith("dec_") and k.endswith(".W") and v.ndim == 4:
                self.Wt[k] = torch.flip(v, (2, 3)).contiguous() if deconv_flip else v

    def bn(self, x, name):
        P = self.P
        shp = (1, -1) + (1,) * (x.ndim - 2)
        return (x - P[name + ".mean"].reshape(shp)) * (P[name + ".gamma"] * P[name + ".inv_std"]).reshape(shp) \
            + P[name + ".beta"].reshape(shp)

    def conv(self, x, name, bias=False):  # IAN_simple.py:73-116
        return F.conv2d(x, self.P[name + ".W"], self.P[name + ".b"] if bias else None, stride=2, padding=2)

    def deconv(self, x, name):  # layers.py:436-483
        return F.conv_transpose2d(x, self.Wt[name + ".W"], None, stride=2, padding=2, output_padding=1)

    def mdcl(self, x, name, scales):  # layers.py:207-258
        P = self.P
        W = P[name + "W"]
        out = F.conv2d(x, W, padding=1) * P[name + "_coeff_base"].reshape(1, -1, 1, 1)
        for s in scales:
            if s == 0:
                out = out + F.conv2d(x, W.mean((2, 3), keepdim=True)) * P[name + "_coeff_1x1"].reshape(1, -1, 1, 1)
            else:
                out = out + F.conv2d(x, W, padding=s, dilation=s) * P[name + "_coeff_%d" % s].reshape(1, -1, 1, 1)
        return out

    def mdblock(self, x, name, scales):  # layers.py:411-416
        a = _act("lrelu", self.bn(x, name + "bnorm0"))
        c = _act("lrelu", self.bn(self.mdcl(a, name, scales), name + "bnorm1"))
        d = self.mdcl(c, name + "2", scales)
        return _act("lrelu", self.bn(x + d, name + "bnorm2"))

    def made(self, z, name):  # layers.py:735-853
        P, (M0, M1, MD) = self.P, self.masks
        h = torch.relu(z @ (P[name + "_input.W"] * M0) + P[name + "_input.b"])
        return (h @ (P[name + "_output_W.W"] * M1) + P[name + "_output_W.b"]) + \
               (z @ (P[name + "_output_D.W"] * MD) + P[name + "_output_D.b"])

    def made_hidden(self, z, name):  # the MaskedLayer that layers.py:775 leaves in MADE.input_layer
        return torch.relu(z @ (self.P[name + "_input.W"] * self.masks[0]) + self.P[name + "_input.b"])

    def features(self, x):
        h1 = _act("lrelu", self.conv(x, "enc_conv1", True))
        h2 = _act("lrelu", self.bn(self.conv(h1, "enc_conv2"), "bnorm2"))
        h3 = _act("lrelu", self.bn(self.conv(h2, "enc_conv3"), "bnorm3"))
        h4 = _act("lrelu", self.bn(self.conv(h3, "enc_conv4"), "bnorm4"))
        return [h1, h2, h3, h4]

    def Zfn(self, x):
        P = self.P
        h4 = self.features(x)[-1]
        fa = "elu" if self.arch == "IAN_simple" else "relu"
        f = _act(fa, self.bn(h4.flatten(1) @ P["enc_fc1.W"], "bnorm_enc_fc1"))
        return self.bn(f @ P["enc_mu.W"], "mu_bnorm")

    def Z_IAF_fn(self, z):
        if self.arch == "IAN_simple":
            return z
        # the reference graph feeds each MADE with its own first masked layer's output (ian_oracle.made_as_wired)
        return (z - self.made(self.made_hidden(z, "l_IAF_mu"), "l_IAF_mu")) / torch.exp(self.made(self.made_hidden(z, "l_IAF_ls"), "l_IAF_ls"))

    def encode(self, x):
        return self.Z_IAF_fn(self.Zfn(x))

    def decode(self, z):
        P = self.P
        if self.arch == "IAN_simple":
            h = _act("relu", self.bn(z @ P["l_dec_fc2.W"], "bnorm_dec_fc2")).reshape(-1, 1024, 4, 4)
            for i in (1, 2, 3):
                h = _act("relu", self.bn(self.deconv(h, "dec_conv%d" % i), "bnorm_dc%d" % i))
            return torch.tanh(self.deconv(h, "dec_out"))
        h = _act("lrelu", z @ P["l_dec_fc2.W"] + P["l_dec_fc2.b"]).reshape(-1, 512, 4, 4)
        for dc, blk, sc in (("dec_conv1", "dec_conv2a", [0, 2]), ("dec_conv2", "dec_conv3a", [0, 2, 3]),
                            ("dec_conv3", "dec_conv4a", [0, 2, 3])):
            h = self.mdblock(self.deconv(h, dc), blk, sc)
        h = _act("lrelu", self.bn(self.deconv(h, "dec_conv4"), "bnorm_dc4"))
        sc = [2, 3, 4]
        R = torch.sigmoid(self.mdcl(h, "R", sc))
        G = torch.sigmoid(self.mdcl(h, "G_a", sc) + self.mdcl(R, "G_b", sc))
        B = torch.sigmoid(self.mdcl(h, "B_a", sc) + self.mdcl(torch.cat([R, G], 1), "B_b", sc))
        beta = lambda t: 2 * (t[:, 0:1] / (t[:, 0:1] + t[:, 1:2] + 1e-8)) - 1
        return torch.cat([beta(R), beta(G), beta(B)], 1)

    # ---- API.py:59,64 gradients ------------------------------------------------
    def imgrad(self, c1, r1, c2, r2, z):
        """API.py:59: grad of mean(X_hat[0,:,r1:r2,c1:c2]) wrt Z."""
        z = torch.as_tensor(np.asarray(z), dtype=self.dtype).clone().requires_grad_(True)
        xh = self.decode(z)
        loss = xh[0, :, r1:r2, c1:c2].mean()
        (g,) = torch.autograd.grad(loss, z)
        return g.numpy()

    def imgradRGB(self, c1, r1, c2, r2, rgb, z):
        """API.py:64: grad of mean(sqr(-X_hat[0,:,r1:r2,c1:c2]+RGB[0,:,r1:r2,c1:c2])) wrt Z."""
        z = torch.as_tensor(np.asarray(z), dtype=self.dtype).clone().requires_grad_(True)
        rgb = torch.as_tensor(np.asarray(rgb), dtype=self.dtype)
        xh = self.decode(z)
        loss = ((-xh[0, :, r1:r2, c1:c2] + rgb[0, :, r1:r2, c1:c2]) ** 2).mean()
        (g,) = torch.autograd.grad(loss, z)
        return g.numpy()

    # numpy in / numpy out conveniences
    def np_encode(self, x):
        with torch.no_grad():
            return self.encode(torch.as_tensor(np.asarray(x), dtype=self.dtype)).numpy()

    def np_decode(self, z):
        with torch.no_grad():
            return self.decode(torch.as_tensor(np.asarray(z), dtype=self.dtype)).numpy()
