import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from oracle import ian_oracle as O
from oracle.train_twin import TrainTwin, make_train_params, ENC_PARAMS
from neural_photo_editor_amd.trainer import Trainer, ENC_WIDTHS
B = 4
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
ref = torch.autograd.grad((-torch.log(T["p_X_hat"][:, 1])).mean(), enc, retain_graph=True)
for i in range(4):
    a_g = nhwc(tr.EH["a%d" % (i + 1)]); a_t = T["g_X_hat"][i].detach().numpy()
    print("layer %d sign flips (gpu X_hat)" % (i + 1), int(((a_g > 0) != (a_t > 0)).sum()), "of", a_t.size)
tr.touched = set(); tr.enc_backward(tr.EH, (1, 1.0 / tr.N, -1, 0.0), False, True, False)
report("EH pass, GPU X_hat   ", tr.grads_numpy("enc"), ref, ENC_PARAMS)
xh = T["X_hat"].detach().numpy().astype(np.float32)
tr.enc_forward(tr.EH, d(xh), targets=(0, 1), acc_target=1)
for i in range(4):
    a_g = nhwc(tr.EH["a%d" % (i + 1)]); a_t = T["g_X_hat"][i].detach().numpy()
    print("layer %d sign flips (twin X_hat)" % (i + 1), int(((a_g > 0) != (a_t > 0)).sum()), "of", a_t.size, "fwd rel", rel(a_g, a_t))
tr.touched = set(); tr.enc_backward(tr.EH, (1, 1.0 / tr.N, -1, 0.0), False, True, False)
report("EH pass, twin X_hat  ", tr.grads_numpy("enc"), ref, ENC_PARAMS)
# same experiment on X with a perturbation of the size of the X_hat discrepancy
ref = torch.autograd.grad(L["discrim_d_loss"], enc, retain_graph=True)
Xp = X + np.random.RandomState(1).uniform(-3e-5, 3e-5, X.shape).astype(np.float32)
tr.enc_forward(tr.EX, d(Xp), targets=(0, -1), acc_target=0)
tr.touched = set(); tr.enc_backward(tr.EX, (0, 1.0 / tr.N, -1, 0.0), False, True, False)
report("EX pass, X + 3e-5 noise", tr.grads_numpy("enc"), ref, ENC_PARAMS)
