"""Run on the GPU box: step-by-step comparison of the training step with the float64 CPU twin."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
from oracle import ian_oracle as O
from oracle.train_twin import TrainTwin, make_train_params
from neural_photo_editor_amd.trainer import Trainer

B = int(os.environ.get("B", "4"))
def rel(a, b): return float(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max() / (np.abs(np.asarray(b)).max() + 1e-30))
def nhwc(t, c): return t.cpu().numpy()[..., :c].transpose(0, 3, 1, 2)
P = make_train_params(O.make_params("IAN", 1))
tr = Trainer(os.path.join(ROOT, "neural_photo_editor_amd", "configs", "IAN.py"), P, batch=B)
tw = TrainTwin(P, dtype=torch.float64)
X = O.make_images(B, seed=1); Z = O.make_latents(B, seed=6); eps = np.random.RandomState(7).randn(B, 100).astype(np.float32)
d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
t0 = time.time(); tr.forward(d(X), d(Z), d(eps)); torch.cuda.synchronize(); print("forward %.2fs" % (time.time() - t0))
L = tw.losses(X, Z, eps); T = tw.tensors
for i in range(4):
    print("g_X[%d]" % i, rel(nhwc(tr.EX["a%d" % (i + 1)], 10**6), T["g_X"][i].detach().numpy()))
print("mu", rel(tr.ZS["mu"].cpu().numpy()[:, :100], T["mu"].detach().numpy()), "ls", rel(tr.ZS["ls"].cpu().numpy()[:, :100], T["ls"].detach().numpy()),
      "z0", rel(tr.ZS["z0"].cpu().numpy()[:, :100], T["z0"].detach().numpy()))
print("X_hat", rel(tr.DZ["xhat"].cpu().numpy(), T["X_hat"].detach().numpy()), "X_gen", rel(tr.DG["xhat"].cpu().numpy(), T["X_gen"].detach().numpy()))
for i in range(4):
    print("g_X_hat[%d]" % i, rel(nhwc(tr.EH["a%d" % (i + 1)], 10**6), T["g_X_hat"][i].detach().numpy()))
print("p_X", rel(tr.EX["p"].cpu().numpy(), T["p_X"].detach().numpy()), "p_X_hat", rel(tr.EH["p"].cpu().numpy(), T["p_X_hat"].detach().numpy()),
      "p_X_gen", rel(tr.EG["p"].cpu().numpy(), T["p_X_gen"].detach().numpy()))
m = tr.metrics()
for k in sorted(m): print("  %-16s gpu %.6f twin %.6f" % (k, m[k], float(L[k])))
g, _ = tw.gradients(X, Z, eps)
for which in ("gen", "discrim"):
    t0 = time.time(); tr.backward(which); tr._regularizers(which); torch.cuda.synchronize(); print(which, "backward %.2fs" % (time.time() - t0))
    rows = []
    for gname in (("dec", "Z") if which == "gen" else ("enc", "Z")):
        got = tr.grads_numpy(gname)
        for name, ref in g[gname].items():
            rows.append((rel(got[name], ref.detach().numpy()), name, float(np.abs(got[name]).max()), float(ref.abs().max())))
    rows.sort(reverse=True)
    for r in rows[:25]: print("   %.3e %-28s |gpu| %.3e |ref| %.3e" % r)
    print("   ... median rel err %.3e over %d tensors" % (np.median([r[0] for r in rows]), len(rows)))
if B >= 16:
    for which in ("gen", "discrim"):
        torch.cuda.synchronize(); t0 = time.time()
        for _ in range(3): tr.step(which, d(X), d(Z), d(eps), return_metrics=False)
        torch.cuda.synchronize(); print("step", which, "%.1f ms" % ((time.time() - t0) / 3 * 1e3))
