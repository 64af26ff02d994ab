"""GPU record (not in the test suite: minutes of float64 CPU work): the decomposition of tests/test_gpu_decomposition.py at the
BENCHMARKED batch, 128 images, update_gen -- where round 3 measured dec_conv4.W 3.8e-2 against the plain float64 twin.
python scripts/exp/decomposition_gpu.py [batch] [which] > gpurun_out/decomposition_b128_gen.json"""
import json, os, sys, time
import numpy as np, torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import ian_oracle as O
from oracle.staged_twin import StagedTwin
from oracle.train_twin import make_train_params
from neural_photo_editor_amd.trainer import Trainer
from test_gpu_decomposition import hip_provider, rel, CFG

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
which = sys.argv[2] if len(sys.argv) > 2 else "gen"
torch.set_num_threads(min(64, os.cpu_count() or 8))
P = make_train_params(O.make_params("IAN", 1))
X, Z = O.make_images(B, seed=31), O.make_latents(B, seed=32)
eps = np.random.RandomState(33).randn(B, 100).astype(np.float32)
tr = Trainer(CFG, P, batch=B)
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
tr.forward(dev(X), dev(Z), dev(eps)); tr.backward(which); tr._finish_allreduce(which); tr._regularizers(which)
torch.cuda.synchronize()
groups = ("dec", "Z") if which == "gen" else ("enc", "Z")
got = {g: tr.grads_numpy(g) for g in groups}
tw = StagedTwin(P, dtype=torch.float64)
t0 = time.time()
plain, _ = tw.gradients_staged(X, Z, eps, which)
at, _ = tw.gradients_staged(X, Z, eps, which, provider=hip_provider(tr))
local = sorted(((v, "%s.%s" % k) for k, v in tw.local_err.items()), reverse=True)
ep = sorted(((rel(got[g][n], plain[g][n].numpy()), n) for g in groups for n in plain[g]), reverse=True)
ea = sorted(((rel(got[g][n], at[g][n].numpy()), n) for g in groups for n in at[g]), reverse=True)
print(json.dumps({"batch": B, "which": which, "twin_seconds": time.time() - t0,
                  "local_forward_error": {"median": float(np.median([e for e, _ in local])), "worst": local[:5]},
                  "grad_vs_plain_float64": {"median": float(np.median([e for e, _ in ep])), "worst": ep[:5]},
                  "grad_vs_float64_at_hip_forward_point": {"median": float(np.median([e for e, _ in ea])), "worst": ea[:5]}}, indent=1))
