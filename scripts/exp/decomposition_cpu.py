"""CPU record of the decomposition (oracle/staged_twin.py) with a float32 run of the restatement as 'the implementation':
gradient error vs the plain float64 twin (forward drift x conditioning) and vs the float64 gradient at the implementation's own
forward point (backward arithmetic only).  python scripts/exp/decomposition_cpu.py > profiles/r04_decomposition_cpu_twin32.json"""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from oracle import ian_oracle as O
from oracle.staged_twin import StagedTwin
from oracle.train_twin import make_train_params

fx = np.load(os.path.join(os.path.dirname(__file__), "..", "..", "tests", "golden", "ref_train_IAN.npz"))
B = int(fx["batch"])
X, Z, eps = fx["X"][:B].astype(np.float32), fx["Z"][:B].astype(np.float32), fx["gen/eps"].astype(np.float32)
P = make_train_params(O.make_params("IAN", 1))
rel = lambda a, b: float(np.abs(a.double().numpy() - b.double().numpy()).max() / (np.abs(b.double().numpy()).max() + 1e-30))
impl, t64 = StagedTwin(P, dtype=torch.float32), StagedTwin(P, dtype=torch.float64)
out = {}
for which in ("gen", "discrim"):
    g32, _ = impl.gradients_staged(X, Z, eps, which)
    rec = dict(impl.rec)
    plain, _ = t64.gradients_staged(X, Z, eps, which)
    at, _ = t64.gradients_staged(X, Z, eps, which, provider=lambda tag, name: rec[(tag, name)])
    ep = sorted(((rel(g32[g][n], plain[g][n]), n) for g in plain for n in plain[g]), reverse=True)
    ea = sorted(((rel(g32[g][n], at[g][n]), n) for g in plain for n in plain[g]), reverse=True)
    le = sorted(((v, "%s.%s" % k) for k, v in t64.local_err.items()), reverse=True)
    out[which] = {"vs_plain_float64": {"median": float(np.median([e for e, _ in ep])), "worst": ep[:4]},
                  "vs_float64_at_own_forward_point": {"median": float(np.median([e for e, _ in ea])), "worst": ea[:4]},
                  "local_forward_error": {"median": float(np.median([e for e, _ in le])), "worst": le[:4]}}
print(json.dumps(out, indent=1))
