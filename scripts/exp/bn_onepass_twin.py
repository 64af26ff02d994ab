"""CPU experiment (round 4, verdict item 1): how far do update_gen's gradients move when the float32 twin computes the
batch-norm variance as E[x^2]-E[x]^2 (what kernels_train.hip did through round 3) instead of Lasagne's two-pass
input.var (minilasagne.py:607)?  Reference = the float64 twin.  python scripts/exp/bn_onepass_twin.py [batch]"""
import os, sys, json
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from oracle.train_twin import TrainTwin, BN_EPS
from oracle import ian_oracle as O
from neural_photo_editor_amd.synthetic import make_train_params

NB = int(sys.argv[1]) if len(sys.argv) > 1 else 16
P = make_train_params(O.make_params("IAN", 1))
X, Z = O.make_images(NB, seed=31), O.make_latents(NB, seed=32)
eps = np.random.RandomState(33).randn(NB, 100).astype(np.float32)


class OnePass(TrainTwin):
    def bn(self, x, name):
        axes = [0] + list(range(2, x.ndim))
        n = float(x.numel() // x.shape[1])
        mean = x.sum(axes, keepdim=True) / n
        var = torch.clamp((x * x).sum(axes, keepdim=True) / n - mean * mean, min=0.0)
        shp = (1, -1) + (1,) * (x.ndim - 2)
        istd = 1.0 / torch.sqrt(var + BN_EPS)
        sc = self.P[name + ".gamma"].reshape(shp) * istd
        return x * sc + (self.P[name + ".beta"].reshape(shp) - mean * sc)


def grads(tw):
    c = tw.cfg
    L = tw.losses(X, Z, eps)
    gen_loss = L["adv_gen"] + c["recon_weight"] * L["pixel_loss"] + c["feature_weight"] * L["feature_loss"] + L["l2_gen"]
    z_loss = c["feature_weight"] * L["feature_loss"] + c["recon_weight"] * L["pixel_loss"] + L["adv_gen"] + L["kl_div"] + L["l2_Z"]
    names = {"dec": list(tw.groups["dec"]), "Z": list(tw.groups["Z"])}
    g_dec = torch.autograd.grad(gen_loss, [tw.P[n] for n in names["dec"]], retain_graph=True)
    g_z = torch.autograd.grad(z_loss, [tw.P[n] for n in names["Z"]])
    return {"dec": dict(zip(names["dec"], [g.double().numpy() for g in g_dec])),
            "Z": dict(zip(names["Z"], [g.double().numpy() for g in g_z]))}, {k: float(v) for k, v in L.items()}


def rel(a, b):
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


torch.set_num_threads(8)
ref, Lr = grads(TrainTwin(P, dtype=torch.float64))
out = {}
for label, cls in (("two_pass32", TrainTwin), ("one_pass32", OnePass)):
    g, L = grads(cls(P, dtype=torch.float32))
    out[label] = {}
    for grp in ("dec", "Z"):
        errs = sorted(((rel(g[grp][n], ref[grp][n]), n) for n in ref[grp]), reverse=True)
        out[label][grp] = {"median": float(np.median([e for e, _ in errs])), "worst": errs[:5]}
    out[label]["loss_err"] = max(abs(L[k] - Lr[k]) / max(1, abs(Lr[k])) for k in Lr if k != "discrim_acc")
print(json.dumps(out, indent=1))
