"""Run on the GPU box: isolate single backward passes of the training step against float64 autograd."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from oracle import ian_oracle as O
from oracle.train_twin import TrainTwin, make_train_params, ENC_PARAMS
from neural_photo_editor_amd.trainer import Trainer, ENC_WIDTHS

B = int(os.environ.get("B", "4"))
def rel(a, b): return float(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max() / (np.abs(np.asarray(b)).max() + 1e-30))
def nhwc(t, c=10**6): return t.cpu().numpy()[..., :c].transpose(0, 3, 1, 2)
P = make_train_params(O.make_params("IAN", 1))
tr = Trainer(os.path.join(ROOT, "neural_photo_editor_amd", "configs", "IAN.py"), P, batch=B)
tw = TrainTwin(P, dtype=torch.float64)
X = O.make_images(B, seed=1); Z = O.make_latents(B, seed=6); eps = np.random.RandomState(7).randn(B, 100).astype(np.float32)
d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
tr.forward(d(X), d(Z), d(eps)); torch.cuda.synchronize()
L = tw.losses(X, Z, eps, stop_xhat=True); T = tw.tensors
enc = [tw.P[n] for n in ENC_PARAMS]
def report(tag, got, ref_list, names):
    rows = sorted(((rel(got[n], r.detach().numpy()), n) for n, r in zip(names, ref_list)), reverse=True)
    print(tag, " ".join("%s=%.1e" % (n, e) for e, n in rows[:8]))
# (1) EX pass only: CE(p_X, 0)
ref = torch.autograd.grad(L["discrim_d_loss"], enc, retain_graph=True)
tr.touched = set(); tr.enc_backward(tr.EX, (0, 1.0 / tr.N, -1, 0.0), False, True, False)
report("EX pass  ", tr.grads_numpy("enc"), ref, ENC_PARAMS)
# intermediate: gradient wrt g_X[3] (a4) and the conv4 pre-BN output
ga = torch.autograd.grad(L["discrim_d_loss"], T["g_X"], retain_graph=True, allow_unused=True)
for i in (3, 2, 1, 0):
    if ga[i] is not None:
        print("   dL/d a%d (value-grad)  twin max %.3e" % (i + 1, float(ga[i].abs().max())))
# (2) EH pass only, X_hat constant: CE(p_X_hat, 1)
pXh = T["p_X_hat"]
ref = torch.autograd.grad((-torch.log(pXh[:, 1])).mean(), enc, retain_graph=True)
tr.touched = set(); tr.enc_backward(tr.EH, (1, 1.0 / tr.N, -1, 0.0), False, True, False)
report("EH pass  ", tr.grads_numpy("enc"), ref, ENC_PARAMS)
# (3) EG pass
pXg = T["p_X_gen"]
ref = torch.autograd.grad((-torch.log(pXg[:, 2])).mean(), enc, retain_graph=True)
tr.touched = set(); tr.enc_backward(tr.EG, (2, 1.0 / tr.N, -1, 0.0), False, True, False)
report("EG pass  ", tr.grads_numpy("enc"), ref, ENC_PARAMS)
# (4) value-gradients into the image for the gen path: d CE(p_X_hat,0) / d X_hat
L2 = tw.losses(X, Z, eps); T2 = tw.tensors
gx = torch.autograd.grad((-torch.log(T2["p_X_hat"][:, 0])).mean(), T2["X_hat"], retain_graph=True)[0]
tr.touched = set(); tr.enc_backward(tr.EH, (0, 1.0 / tr.N, -1, 0.0), False, False, True)
got = tr.EH["dx"].cpu().numpy()[..., :3].transpose(0, 3, 1, 2)
print("d CE(p_X_hat,0)/d X_hat", rel(got, gx.numpy()), "max", float(gx.abs().max()))
# (5) feature-loss only into X_hat
gx = torch.autograd.grad(L2["feature_loss"], T2["X_hat"], retain_graph=True)[0]
for i, w in enumerate(ENC_WIDTHS):
    cnt = (32 >> i) ** 2 * w
    tr.k.pair_loss(tr.EH["a%d" % (i + 1)], tr.EX["a%d" % (i + 1)], tr.EH["da%d" % (i + 1)], B * cnt, 1, 1, 1, 1.0 / (4.0 * tr.N * cnt), 0, tr.ws_loss, 1024, 0.0, tr.scalars[40:42])
tr.enc_backward(tr.EH, (-1, 0.0, -1, 0.0), True, False, True)
got = tr.EH["dx"].cpu().numpy()[..., :3].transpose(0, 3, 1, 2)
print("d feature_loss/d X_hat", rel(got, gx.numpy()), "max", float(gx.abs().max()))
# (6) decoder backward from a random image gradient
rs = np.random.RandomState(0); gimg = rs.randn(B, 3, 64, 64).astype(np.float32) * 1e-3
from oracle.train_twin import decoder_param_names
dn = decoder_param_names()
ref = torch.autograd.grad((T2["X_hat"] * torch.tensor(gimg, dtype=torch.float64)).sum(), [tw.P[n] for n in dn] + [T2["z0"]], retain_graph=True)
tr.touched = set(); tr.DZ["dxhat"].copy_(d(gimg)); tr.dec_backward(tr.DZ, tr.ZS["z"], True, True)
report("DZ pass  ", tr.grads_numpy("dec"), ref[:-1], dn)
