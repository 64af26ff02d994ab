"""CPU experiment (round 4, verdict item 1): how far do update_gen's float32 gradients move when every linear layer's output
carries a relative rounding perturbation of k float32 ulps (k = 1: a correctly rounded blocked sum; k ~ 4-8: a sequential
K = 3200..12800 accumulation chain, what one MFMA accumulator does)?  Reference = the float64 twin.  Shows the CONDITIONING
of the composed gradient test: the spread of the worst-tensor error over perturbation seeds is the bar any float32
implementation can be held to.   python scripts/exp/fp32_noise_conditioning.py [batch] [k ...]"""
import os, sys, json
import numpy as np, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from oracle.train_twin import TrainTwin
from oracle import ian_oracle as O
from neural_photo_editor_amd.synthetic import make_train_params

NB = int(sys.argv[1]) if len(sys.argv) > 1 else 16
KS = [float(a) for a in sys.argv[2:]] or [0.0, 1.0, 4.0]
P = make_train_params(O.make_params("IAN", 1))
X, Z = O.make_images(NB, seed=31), O.make_latents(NB, seed=32)
eps = np.random.RandomState(33).randn(NB, 100).astype(np.float32)
EPS32 = 2.0 ** -24


class Noisy(TrainTwin):
    k, gen = 0.0, None

    def _n(self, t):
        if self.k == 0.0:
            return t
        u = torch.empty_like(t).uniform_(-1, 1, generator=self.gen)
        return t * (1 + self.k * EPS32 * u)

    def deconv(self, x, name):
        return self._n(TrainTwin.deconv(self, x, name))

    def mdcl(self, x, name, scales):
        return self._n(TrainTwin.mdcl(self, x, name, scales))

    def bn(self, x, name):
        return TrainTwin.bn(self, self._n(x), name)   # every batch-normalised tensor is a conv / dense output


def grads(tw):
    c = tw.cfg
    L = tw.losses(X, Z, eps)
    gen_loss = L["adv_gen"] + c["recon_weight"] * L["pixel_loss"] + c["feature_weight"] * L["feature_loss"] + L["l2_gen"]
    z_loss = c["feature_weight"] * L["feature_loss"] + c["recon_weight"] * L["pixel_loss"] + L["adv_gen"] + L["kl_div"] + L["l2_Z"]
    names = {"dec": list(tw.groups["dec"]), "Z": list(tw.groups["Z"])}
    g_dec = torch.autograd.grad(gen_loss, [tw.P[n] for n in names["dec"]], retain_graph=True)
    g_z = torch.autograd.grad(z_loss, [tw.P[n] for n in names["Z"]])
    return {"dec": dict(zip(names["dec"], [g.double().numpy() for g in g_dec])),
            "Z": dict(zip(names["Z"], [g.double().numpy() for g in g_z]))}


def rel(a, b):
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


torch.set_num_threads(8)
ref = grads(TrainTwin(P, dtype=torch.float64))
rows = []
for k in KS:
    for seed in range(1 if k == 0 else 3):
        tw = Noisy(P, dtype=torch.float32)
        tw.k, tw.gen = k, torch.Generator().manual_seed(seed)
        g = grads(tw)
        r = {"k_ulps": k, "seed": seed}
        for grp in ("dec", "Z"):
            errs = sorted(((rel(g[grp][n], ref[grp][n]), n) for n in ref[grp]), reverse=True)
            r[grp] = {"median": float(np.median([e for e, _ in errs])), "worst": errs[0]}
        rows.append(r)
        print(json.dumps(r), flush=True)
