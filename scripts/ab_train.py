"""Within-process A/B of one ian_trainer option on the training step (128 images, ian_train_step):
python scripts/ab_train.py key=valueA,valueB [batch]  -> ms per (update_gen + update_discrim) pair, alternated 3 times."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from neural_photo_editor_amd import synthetic as O
from neural_photo_editor_amd.trainer import Trainer
key, vals = sys.argv[1].split("=")
va, vb = [float(v) for v in vals.split(",")]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 128
tr = Trainer(os.path.join(ROOT, "neural_photo_editor_amd", "configs", "IAN.py"), O.make_train_params(O.make_params("IAN", 1)), B)
rs = np.random.RandomState(0)
X = torch.from_numpy(O.make_images(B, seed=1)).cuda()
Z = torch.from_numpy(rs.randn(B, 100).astype(np.float32)).cuda()
eps = torch.from_numpy(rs.randn(B, 100).astype(np.float32)).cuda()
tr.autotune()
res = {va: [], vb: []}
for rep in range(3):
    for v in (va, vb):
        tr.set_option(key, v)
        for w in ("gen", "discrim"):
            tr.step(w, X, Z, eps, return_metrics=False)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(3):
            tr.step("gen", X, Z, eps, return_metrics=False)
            tr.step("discrim", X, Z, eps, return_metrics=False)
        torch.cuda.synchronize()
        res[v].append((time.perf_counter() - t) / 3 * 1e3)
for v in (va, vb):
    print("%s=%g: %s ms per G+D pair (median %.2f) -> %.0f images/s" % (key, v, " ".join("%.2f" % t for t in res[v]), float(np.median(res[v])),
                                                                       2 * B / (float(np.median(res[v])) * 1e-3)))
